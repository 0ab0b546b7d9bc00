#!/bin/bash
# Round 5, eleventh GPU session: hand-off length and wave priorities of the four-wave kernel again, now that it waits more than it issues
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/ab_bench.sh "base x4 x16 p3210 p2100" 3 --no-realtime --no-configs --steps 20 --warmup 5 --repeats 16
bash scripts/ab_bench.sh "base x4 x16" 1 --no-realtime --no-configs
