#!/bin/bash
# Round 5, session "u": the four-wave cut once more, now that the filter wave is light -- (2,6,11) [shipped] against (2,6,8)
# (the filter's envelope, gain and add move to the filter wave) and (2,6,9), interleaved, on the coherent score (188 blocks), the
# incoherent one (the driver's 20 blocks; 94 blocks) and the variant whose cutoff moves every frame.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python -c "import torch" 2>/dev/null
for r in 1 2 3; do
  bash scripts/ab_bench.sh "base c268 c269" 1 --steps 20 --warmup 5 --no-realtime --no-configs | sed "s/^/driver20 /"
  bash scripts/ab_bench.sh "base c268 c269" 1 --no-realtime --no-configs | sed "s/^/steps94 /"
  bash scripts/ab_bench.sh "base c268 c269" 1 --steps 188 --warmup 8 --no-realtime --no-configs | sed "s/^/steps188 /"
  bash scripts/ab_bench.sh "base c268 c269" 1 --steps 188 --warmup 8 --variant survey2 --no-realtime --no-configs | sed "s/^/survey2 /"
done
