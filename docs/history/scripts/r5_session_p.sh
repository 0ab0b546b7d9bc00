#!/bin/bash
# Round 5, session "p": time per block against blocks per launch (fm_voice, 65 536 voices, 188-block regions) -- is there a
# fixed cost per launch behind the gap between the driver's 20-block regions and the 188-block default run?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python -c "import torch" 2>/dev/null
for rep in 1 2; do
for b in 32 4 8 12 16 20 24 32; do
  timeout 300 python bench.py --no-cpu-baseline --no-realtime --no-configs --bus-batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']; t = d['timing']
    cyc = [a * c * 1e3 / 188 for a, c in zip(t['regions_ms'], t['kernel_sclk_ghz'])]
    print('batch $b', 'value %.4g' % d['value'], 'blocks/launch %.2f' % r['blocks_per_launch'], 'kernel_ms_avg %.4f' % r['kernel_ms_avg'], 'launches', r['kernel_launches'],
          'k-cycles/block by region', [round(x, 1) for x in cyc], 'sclk', t['kernel_sclk_ghz'])
"
done; done
