#!/bin/bash
# One gpurun session: GPU test-suite, bench lines, RCCL smoke, rocprofv3 profiles.  Logs under gpurun_out/<tag>/.
# usage: scripts/gpu_session.sh <tag> [parts...]   parts: tests fullsize bench nccl prof (default: all)
TAG=${1:-s}
shift || true
PARTS=${*:-"tests bench nccl prof"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for p in $PARTS; do
  case $p in
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log ;;
    fullsize)
      timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -s > $OUT/fullsize.log 2>&1; echo "rc=$?" >> $OUT/fullsize.log ;;
    bench)
      timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
      timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
      timeout 300 python bench.py --steps 20 --warmup 5 --sparse-events --no-cpu-baseline > $OUT/bench_driver_sparse.json 2> $OUT/bench_driver_sparse.err
      for v in 262144 1048576; do
        timeout 300 python bench.py --voices-per-gpu $v --no-cpu-baseline > $OUT/bench_fm_$v.json 2> $OUT/bench_fm_$v.err
      done
      timeout 300 python bench.py --graph epiano_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline > $OUT/bench_epiano.json 2> $OUT/bench_epiano.err
      timeout 300 python bench.py --graph sat4x_voice --voices-per-gpu 131072 --steps 94 --no-cpu-baseline > $OUT/bench_sat4x.json 2> $OUT/bench_sat4x.err
      timeout 300 python bench.py --graph sub_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline > $OUT/bench_sub.json 2> $OUT/bench_sub.err
      timeout 300 python bench.py --midi-live 1000 --no-cpu-baseline > $OUT/bench_midi_live.json 2> $OUT/bench_midi_live.err
      ;;
    nccl)
      # RCCL refuses two ranks on one GPU: (a) the RCCL leg with a one-rank communicator, (b) the self-launching
      # two-rank path on one device with the reduce over gloo
      timeout 300 python bench.py --dist-single --steps 20 --warmup 4 --no-cpu-baseline > $OUT/rccl1.json 2> $OUT/rccl1.err; echo "rc=$?" >> $OUT/rccl1.err
      timeout 300 python bench.py --gpus 2 --single-device --backend gloo --steps 20 --warmup 4 --voices-per-gpu 32768 --no-cpu-baseline > $OUT/gloo2.json 2> $OUT/gloo2.err; echo "rc=$?" >> $OUT/gloo2.err ;;
    dbg)
      timeout 600 python scripts/dbg_fullsize.py > $OUT/dbg.log 2>&1 ;;
    cluster)
      timeout 600 python -m pytest tests/test_cluster_gpu.py -m gpu -q > $OUT/cluster.log 2>&1 ;;
    tests_all)
      timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log ;;
    prof)
      # (the command the driver's BENCH line is taken with: its blocks-per-launch must match for the PMC numbers to apply)
      PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536 --steps 20 --warmup 5
      PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_default
      PROF_SUMMARY_ARGS="262144 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm262144 --voices-per-gpu 262144
      PROF_SUMMARY_ARGS="262144 256 epiano_voice" bash scripts/gpu_profile.sh ${TAG}_epiano --graph epiano_voice --voices-per-gpu 262144 --steps 94
      PROF_SUMMARY_ARGS="131072 256 sat4x_voice" bash scripts/gpu_profile.sh ${TAG}_sat4x --graph sat4x_voice --voices-per-gpu 131072 --steps 94
      ;;
  esac
done
ls -la $OUT
tail -5 $OUT/*.log 2>/dev/null
for f in $OUT/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print("value %.4g ms/step %.4f kern_ms %.4f frac %.4f hash %s variant %s stale %s rccl %s ev %s" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("kernel_hash"), r.get("kernel_variant"), r.get("stale_profile"), d.get("rccl_ranks"), d["config"].get("events_in_timed_region")))
    if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["whole_bank_per_thread"]["value"])
except Exception as e:
    print("unreadable", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
