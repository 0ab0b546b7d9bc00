#!/bin/bash
# round 4, session 4: which wave handles the events -- pipeline shapes A/B at the driver's command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04d; mkdir -p $OUT
timeout 600 python -m pytest tests/test_plugin_gpu.py tests/test_parity_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
bash scripts/ab_bench.sh "base b4 kahn kahn4 k3e kahn5" 3 --no-realtime --steps 20 --warmup 5 > $OUT/ab_driver.log 2>&1
cat $OUT/ab_driver.log
bash scripts/ab_bench.sh "base b4 kahn4 k3e" 2 --no-realtime > $OUT/ab_default.log 2>&1
cat $OUT/ab_default.log
