#!/bin/bash
# round 4, session 9: sticky chunk loops (OGC_STICKY=1) -- A/B and parity of the variant library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04s; mkdir -p $OUT
bash scripts/ab_bench.sh "base st" 2 --no-realtime > $OUT/ab_94.log 2>&1
bash scripts/ab_bench.sh "base st" 2 --no-realtime --steps 20 --warmup 5 > $OUT/ab_driver.log 2>&1
bash scripts/ab_bench.sh "base st" 1 --no-realtime --voices-per-gpu 131072 > $OUT/ab_131k.log 2>&1
cat $OUT/ab_94.log $OUT/ab_driver.log $OUT/ab_131k.log
export OGC_STICKY=1 OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_st.so
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_jit_gpu.py tests/test_codegen_fuzz_gpu.py tests/test_event_edges_gpu.py -m gpu -q -x -k "not fm262144 and not fm1048576 and not 8388608" > $OUT/pytest_st.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_st.log
tail -5 $OUT/pytest_st.log
