import sys, json
tag = sys.argv[1]
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]
    print(tag, "value %.4g kernel_ms %.4f ms/step %.4f depth %s" % (d["value"], r["kernel_ms_avg"], d["ms_per_step"], r.get("pipeline_waves_per_64_voices")))
