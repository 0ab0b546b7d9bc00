#!/bin/bash
# round 4, session 5: depth sweep by bank size (calibrates the engine's depth model), suite, config lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT
for v in 16384 32768 49152 65536 98304 131072 196608 262144; do
  for s in 0 2 4 auto; do
    if [ $s = auto ]; then unset OSCEN_GPU_SPLIT; else export OSCEN_GPU_SPLIT=$s; fi
    python bench.py --voices-per-gpu $v --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py "v$v split$s"
  done
done > $OUT/sweep.log 2>&1
unset OSCEN_GPU_SPLIT
cat $OUT/sweep.log
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 300 python bench.py --graph epiano_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py epiano
timeout 300 python bench.py --graph sat4x_voice --voices-per-gpu 131072 --steps 94 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py sat4x
timeout 300 python bench.py --graph sub_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py sub
timeout 300 python bench.py --voices-per-gpu 1048576 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py fm1M
