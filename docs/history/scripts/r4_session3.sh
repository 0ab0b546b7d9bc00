#!/bin/bash
# round 4, session 3: float-countdown ADSR + tick contraction: suite, PMC profile of the driver's command, region trend, A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04c; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 --repeats 80 --no-realtime --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['timing']['regions_ms']
print('regions trend (ms):', ' '.join('%.3f'%x for x in r[:12]), '...', ' '.join('%.3f'%x for x in r[-6:]), 'median value %.4g'%d['value'])
" > $OUT/trend.log 2>&1
cat $OUT/trend.log
bash scripts/ab_bench.sh "base cut5 kahn nosync" 3 --no-realtime --steps 20 --warmup 5 > $OUT/ab_driver.log 2>&1
for s in 4 2; do OSCEN_GPU_SPLIT=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py split$s; done >> $OUT/ab_driver.log 2>&1
cat $OUT/ab_driver.log
PROF_FULL=1 PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh r04c_fm65536 --steps 20 --warmup 5 > $OUT/prof.log 2>&1
tail -5 $OUT/prof.log; cat gpurun_out/profiles_out/r04c_fm65536_summary.md 2>/dev/null | head -60
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04c/bench_driver.json").read().strip().splitlines()[-1])
print("value %.4g" % d["value"], "rt", d.get("realtime_voices_at_48k"), d["timing"]["regions_ms"])
rt=d["realtime"]
for r in rt["loaded"]["runs"]:
    print("loaded", r["voices"], r["blocks"], {k: round(v,3) for k,v in r["latency_ms"].items()}, "miss", r["deadline_misses"], "setup_s %.1f first_ms %.1f" % (r["setup_s"], r["first_block_ms"]), r["event_stats"], r.get("note"), r.get("paced"), r["worst_blocks"][:3])
print("paced", rt["loaded"]["paced"])
for r in rt["idle_bank"]["runs"]:
    print("idle", r["voices"], round(r["latency_ms"]["p50"],3), round(r["latency_ms"]["p99"],3), round(r["latency_ms"]["max"],3), r["deadline_misses"])
PY
