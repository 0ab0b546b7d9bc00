#!/bin/bash
# round 4, session 1: GPU suite on the tolerance-mode kernels, bench lines, interleaved A/B against -DOG_STRICT
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04a; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
bash scripts/ab_bench.sh "base strict" 3 --no-realtime > $OUT/ab_strict.log 2>&1
bash scripts/ab_bench.sh "base strict" 3 --no-realtime --steps 20 --warmup 5 > $OUT/ab_strict_driver.log 2>&1
tail -8 $OUT/pytest.log; cat $OUT/ab_strict.log $OUT/ab_strict_driver.log
for f in bench_driver bench_default; do python scripts/benchline.py $f < $OUT/$f.json; done
