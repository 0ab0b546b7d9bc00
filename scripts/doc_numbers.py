"""Print the figures the documents quote (README.md) from a bench_<tag>_configs.json.

    python scripts/doc_numbers.py profiles/bench_r03g_configs.json
"""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/bench_r03g_configs.json"))
for k, v in d.items():
    r = v["roofline"]
    print("%-14s value %.3e  ms/step %.4f  kernel %.1f us/block (%s, %.1f blocks/launch)  charged %.0f GB/s (%.1f %%)  dram %s  valu %s" % (
        k, v["value"], v["ms_per_step"], r["kernel_ms_per_block"] * 1e3, r["kernel_variant"], r["blocks_per_launch"], r["charged_gbs"],
        100 * r["frac"], ("%.0f GB/s" % r["dram_gbs"]) if r.get("dram_gbs") else "-",
        ("%.0f %% of nominal, %.0f %% of ceiling, %.1f VALU/frame" % (100 * r["valu_issue"]["frac"], 100 * r["valu_issue"]["frac_of_measured_ceiling"],
                                                                 r["valu_issue"]["valu_wave_inst_per_64_voices_per_frame"])) if r.get("valu_issue") else "-"))
drv = d.get("driver") or next(iter(d.values()))
t = drv.get("timing", {})
print("driver regions ms:", t.get("regions_ms"))
rt = drv.get("realtime")
if rt:
    print("realtime_voices_at_48k", rt["realtime_voices_at_48k"], "harness", rt.get("harness"))
    for r in rt["idle_bank"]["runs"]:
        l = r["latency_ms"]
        p = r.get("paced")
        print("  %9d voices: p50 %.3f p99 %.3f p999 %.3f max %.3f ms, misses %d, marker timeouts %d%s" % (
            r["voices"], l["p50"], l["p99"], l["p999"], l["max"], r["deadline_misses"], r["marker_timeouts"],
            "" if not p else "; paced: p50 %.3f p99 %.3f max %.3f ms, misses %d" % (p["latency_ms"]["p50"], p["latency_ms"]["p99"], p["latency_ms"]["max"], p["deadline_misses"])))
    c = rt["cluster_entry_host_side"]["latency_ms"]
    print("  cluster entry host side: p50 %.3f p99 %.3f max %.3f ms" % (c["p50"], c["p99"], c["max"]))
cb = drv.get("cpu_baseline")
if cb:
    print("cpu_baseline %.3e at %d threads; single thread %.3e; criterion %s" % (cb["value"], cb["cores"], cb.get("single_thread", {}).get("value", float("nan")) if isinstance(cb.get("single_thread"), dict) else float("nan"), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in cb.get("criterion_shapes", {}).items() if k != "source"}))
