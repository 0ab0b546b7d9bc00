#!/bin/bash
# Round 5, ninth GPU session: the cut the crossing-aware DP picks ({env3 op3}|{xf env2 op2 mix}|{env1 op1 env_f gain add}|{filter ..}: base)
# against the best measured cut of session 6 (c3611) and round 4's (c369)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/ab_bench.sh "base c3611 c369" 4 --no-realtime --no-configs --steps 20 --warmup 5 --repeats 16
bash scripts/ab_bench.sh "base c3611 c369" 2 --no-realtime --no-configs
bash scripts/ab_bench.sh "base c3611" 1 --no-realtime --no-configs --voices-per-gpu 98304
