#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04g; mkdir -p $OUT
bash scripts/ab_bench.sh "base k5 evu" 3 --no-realtime --steps 20 --warmup 5 > $OUT/ab_driver.log 2>&1
cat $OUT/ab_driver.log
bash scripts/ab_bench.sh "base k5 evu" 2 --no-realtime > $OUT/ab_default.log 2>&1
cat $OUT/ab_default.log
OSCEN_GPU_LIB=$PWD/oscen_amd/_build/liboscen_gpu_k5.so OGC_ALAP=0 OGC_CUTS=1,5,6,8 OSCEN_GPU_SPLIT=4 timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -2
