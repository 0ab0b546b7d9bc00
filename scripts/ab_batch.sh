#!/bin/bash
# driver's command against blocks per launch: is one 20-block launch the best way to render 20 blocks?
for r in 1 2 3; do for b in 0 10 7 5 4; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --bus-batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('batch $b', 'round $r', 'value %.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'ms_per_step %.4f' % d['ms_per_step'], d['roofline'].get('blocks_per_launch'))
"; done; done
