#!/bin/bash
# Round 5, final measurement session (after the LAST kernel change): suite with the observed-error record, the driver's
# whole line, counter profiles of every configuration bench.py prints.
# usage: scripts/r5_final.sh <tag> [parts: tests bench prof]
TAG=${1:-r05f}; shift || true
PARTS=${*:-"tests bench prof"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for p in $PARTS; do case $p in
tests)
  rm -f $OUT/observed.jsonl $OUT/observed.jsonl.hard_regime_curve.json
  OSCEN_OBSERVED=$OUT/observed.jsonl timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  grep -E "passed|failed|FAILED|rc=" $OUT/pytest.log | tail -6
  python scripts/observed_errors.py $OUT/observed.jsonl > $OUT/observed_errors.md 2>&1; head -10 $OUT/observed_errors.md; tail -1 $OUT/observed_errors.md ;;
bench)
  ( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
  timeout 600 python bench.py --no-realtime --no-configs > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 300 python bench.py --voices-per-gpu 1048576 --no-cpu-baseline --no-realtime --no-configs > $OUT/bench_fm_1048576.json 2> /dev/null
  timeout 300 python bench.py --midi-live 1000 --no-cpu-baseline --no-realtime --no-configs > $OUT/bench_midi_live.json 2> /dev/null
  timeout 300 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --no-configs > $OUT/bench_rccl1.json 2> /dev/null
  timeout 300 python bench.py --graph echo_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime --no-configs > $OUT/bench_echo.json 2> /dev/null
  for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; t=d.get("timing",{})
    v=r.get("valu_issue") or {}
    print(os.path.basename(sys.argv[1]), "value %.4g ms/step %.4f kern_ms/block %.5f frac %.4f dram %s valu %s hash %s stale %s" % (d["value"], d["ms_per_step"], r["kernel_ms_per_block"], r["frac"], r.get("dram_gbs"), v.get("frac"), r.get("kernel_variant"), r.get("stale_profile")))
    if t.get("kernel_sclk_ghz"): print("   first5 %.4g" % t["value_median_first5"], "kernel sclk", t["kernel_sclk_ghz"][:3], "..", t["kernel_sclk_ghz"][-3:])
    if d.get("configs"):
        for c in d["roofline"]["configs"]: print("   cfg", {k: c.get(k) for k in ("config","value","ms_per_step","kernel","roofline_frac","valu_issue_frac","dram_gbs","stale_profile","error")})
    if d.get("cpu_baseline"): print("   cpu", "%.3g" % d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["cores_is"])
    rt=d.get("realtime")
    if rt:
        print("   realtime_voices_at_48k", rt["realtime_voices_at_48k"], "idle-bank", rt["idle_bank"]["realtime_voices_at_48k"])
        for r2 in rt["loaded"]["runs"]:
            print("   loaded", r2["voices"], r2["blocks"], {k: round(v,3) for k,v in r2["latency_ms"].items()}, "miss", r2["deadline_misses"], r2.get("note"))
        print("   loaded paced", rt["loaded"]["paced"])
        for r2 in rt["idle_bank"]["runs"]:
            print("   idle", r2["voices"], round(r2["latency_ms"]["p50"],3), round(r2["latency_ms"]["p99"],3), round(r2["latency_ms"]["max"],3), r2["deadline_misses"], r2.get("paced",{}).get("latency_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
  done ;;
prof)
  PROF_FULL=1 PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536 --steps 20 --warmup 5
  PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_default
  PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_survey2 --variant survey2
  PROF_SUMMARY_ARGS="262144 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm262144 --voices-per-gpu 262144
  PROF_SUMMARY_ARGS="1048576 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm1048576 --voices-per-gpu 1048576
  PROF_SUMMARY_ARGS="262144 256 epiano_voice" bash scripts/gpu_profile.sh ${TAG}_epiano --graph epiano_voice --voices-per-gpu 262144 --steps 94
  PROF_SUMMARY_ARGS="262144 256 sub_voice" bash scripts/gpu_profile.sh ${TAG}_sub --graph sub_voice --voices-per-gpu 262144 --steps 94
  PROF_SUMMARY_ARGS="131072 256 sat4x_voice" bash scripts/gpu_profile.sh ${TAG}_sat4x --graph sat4x_voice --voices-per-gpu 131072 --steps 94
  ls $ROOT/gpurun_out/profiles_out/ ;;
esac; done
