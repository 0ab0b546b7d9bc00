#!/bin/bash
# A/B of voices-per-wave: scripts/ab_lanes.sh "64 32 16" rounds [bench args]
L=${1:-"64 32 16"}; ROUNDS=${2:-2}; shift; shift
for r in $(seq 1 $ROUNDS); do for l in $L; do
  OSCEN_GPU_LANES=$l python bench.py --steps 94 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    print('lanes $l round $r', 'value %.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'ms_per_step %.4f' % d['ms_per_step'])
"
done; done
