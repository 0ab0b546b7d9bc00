#!/bin/bash
# Round 5, seventh GPU session: the flattened lazy TPT test against the per-frame test (tl0) where the cutoff moves every
# frame (--variant survey2) and where it does not; the new cut (3,7,11 by the re-weighted DP).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/ab_bench.sh "base tl0" 3 --no-realtime --no-configs --variant survey2
bash scripts/ab_bench.sh "base tl0" 3 --no-realtime --no-configs --steps 20 --warmup 5 --repeats 16
bash scripts/ab_bench.sh "base tl0" 2 --no-realtime --no-configs --voices-per-gpu 262144
bash scripts/ab_bench.sh "base tl0" 1 --no-realtime --no-configs --graph sub_voice --voices-per-gpu 262144
