#!/bin/bash
# Round 5, tenth GPU session: with the instruction count halved the four-wave kernel waits 54 % of its wave cycles
# (profiles/r05f_fm65536_summary.md) -- do five or six waves per 64 voices (more waves per SIMD, shorter stages) pay?
# w5 = OGC_CUTS 2,5,8,11; w6 = 2,5,7,9,11; w6b = 2,4,6,8,11
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/ab_bench.sh "base w5 w6 w6b" 3 --no-realtime --no-configs --steps 20 --warmup 5 --repeats 16
bash scripts/ab_bench.sh "base w5 w6 w6b" 1 --no-realtime --no-configs
bash scripts/ab_bench.sh "base w5 w6" 1 --no-realtime --no-configs --voices-per-gpu 32768
for t in w5 w6; do
  echo "-- parity: $t"
  ( set -a; . oscen_amd/_build/liboscen_gpu_$t.env; set +a; OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_$t.so timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_hard_regime_gpu.py -m gpu -q -k "every_pipeline_depth or default_params or one_second or fast_envelopes or event_edge_cases or variant" 2>&1 | tail -3 )
done
