#!/bin/bash
# round 4, final measurement session: suite with the observed-error record, every bench line, profiles of the shipped kernels
# usage: scripts/r4_final.sh <tag> [parts: tests bench prof rt]
TAG=${1:-r04h}; shift || true
PARTS=${*:-"tests bench prof rt"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for p in $PARTS; do case $p in
tests)
  rm -f $OUT/observed.jsonl
  OSCEN_OBSERVED=$OUT/observed.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -4 $OUT/pytest.log
  python scripts/observed_errors.py $OUT/observed.jsonl > $OUT/observed_errors.md 2>&1; head -12 $OUT/observed_errors.md; tail -1 $OUT/observed_errors.md ;;
bench)
  ( time timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
  timeout 600 python bench.py --no-realtime > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 300 python bench.py --steps 20 --warmup 5 --sparse-events --no-cpu-baseline --no-realtime > $OUT/bench_driver_sparse.json 2> /dev/null
  for v in 262144 1048576; do timeout 300 python bench.py --voices-per-gpu $v --no-cpu-baseline --no-realtime > $OUT/bench_fm_$v.json 2> /dev/null; done
  timeout 300 python bench.py --graph epiano_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime > $OUT/bench_epiano.json 2> /dev/null
  timeout 300 python bench.py --graph sat4x_voice --voices-per-gpu 131072 --steps 94 --no-cpu-baseline --no-realtime > $OUT/bench_sat4x.json 2> /dev/null
  timeout 300 python bench.py --graph sub_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime > $OUT/bench_sub.json 2> /dev/null
  timeout 300 python bench.py --midi-live 1000 --no-cpu-baseline --no-realtime > $OUT/bench_midi_live.json 2> /dev/null
  timeout 300 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-realtime > $OUT/bench_rccl1.json 2> /dev/null
  for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(os.path.basename(sys.argv[1]), "value %.4g ms/step %.4f kern_ms %.4f frac %.4f hash %s variant %s stale %s" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("kernel_hash"), r.get("kernel_variant"), r.get("stale_profile")))
    if d.get("cpu_baseline"): print("   cpu", "%.3g" % d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["cores_is"])
    if d.get("multi_gpu"): print("   multi_gpu", d["multi_gpu"])
    rt=d.get("realtime")
    if rt:
        print("   realtime_voices_at_48k", rt["realtime_voices_at_48k"], "idle-bank", rt["idle_bank"]["realtime_voices_at_48k"])
        for r2 in rt["loaded"]["runs"]:
            print("   loaded", r2["voices"], r2["blocks"], {k: round(v,3) for k,v in r2["latency_ms"].items()}, "miss", r2["deadline_misses"], "worst", [(a, round(b,2)) for a,b in r2["worst_blocks"][:3]], r2["event_stats"]["full_rebuilds"], r2.get("note"))
        print("   loaded paced", rt["loaded"]["paced"])
        for r2 in rt["idle_bank"]["runs"]:
            print("   idle", r2["voices"], round(r2["latency_ms"]["p50"],3), round(r2["latency_ms"]["p99"],3), round(r2["latency_ms"]["max"],3), r2["deadline_misses"], r2.get("paced",{}).get("latency_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
  done ;;
prof)
  PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536 --steps 20 --warmup 5
  PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_default
  PROF_SUMMARY_ARGS="262144 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm262144 --voices-per-gpu 262144
  PROF_SUMMARY_ARGS="262144 256 epiano_voice" bash scripts/gpu_profile.sh ${TAG}_epiano --graph epiano_voice --voices-per-gpu 262144 --steps 94
  PROF_SUMMARY_ARGS="131072 256 sat4x_voice" bash scripts/gpu_profile.sh ${TAG}_sat4x --graph sat4x_voice --voices-per-gpu 131072 --steps 94 ;;
rt)
  bash scripts/prof_realtime.sh ${TAG}_rt8m_loaded 8388608 300 loaded | tail -12
  bash scripts/prof_realtime.sh ${TAG}_rt6m_loaded 6291456 300 loaded | tail -12 ;;
esac; done
ls gpurun_out/profiles_out | grep $TAG
