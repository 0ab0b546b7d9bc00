#!/bin/bash
# Round 5, fifth GPU session: pipeline depth by bank size on the round-5 kernels (the engine's depth model was calibrated on
# round 4's: profiles/r04_depth_sweep.md)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
echo "| voices | depth 1 | depth 2 | depth 4 | auto |"
echo "|---|---|---|---|---|"
for V in 16384 32768 49152 65536 98304 131072 196608 262144; do
  row="| $V |"
  for D in 0 2 4 auto; do
    if [ "$D" = "auto" ]; then unset OSCEN_GPU_SPLIT; else export OSCEN_GPU_SPLIT=$D; fi
    r=$(python bench.py --voices-per-gpu $V --steps 40 --warmup 5 --no-cpu-baseline --no-realtime --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%.4f ms, %.3g (%s)' % (d['ms_per_step'], d['value'], d['roofline']['kernel_variant'].split('_')[1]))
")
    row="$row $r |"
  done
  echo "$row"
done
