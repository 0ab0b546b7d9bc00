#!/bin/bash
# Round 5, sixth GPU session: the lazy TPT parameter test (tl0 = without it) and the cut of the four-wave pipeline now that the
# filter wave is lighter (c<a><b><c> = OGC_CUTS: one-past-last stage of waves 0..2; the shipped cut is 3,6,9).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
echo "== driver's command, 16 regions"
bash scripts/ab_bench.sh "base tl0 c3611 c3711 c4711 c379 c4811" 3 --no-realtime --no-configs --steps 20 --warmup 5 --repeats 16
echo "== 188-block regions"
bash scripts/ab_bench.sh "base tl0 c3611 c3711 c4711" 1 --no-realtime --no-configs
echo "== variant survey2 (the cutoff moves)"
bash scripts/ab_bench.sh "base tl0" 2 --no-realtime --no-configs --variant survey2
echo "== ordinary kernel, 262 144 voices"
bash scripts/ab_bench.sh "base tl0" 2 --no-realtime --no-configs --voices-per-gpu 262144
