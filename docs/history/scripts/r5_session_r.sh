#!/bin/bash
# Round 5, session "r": a region of K blocks between two synchronisations, K = 8 .. 94 (fm_voice, 65 536 voices): cycles per
# block of the region and of the voice-kernel launches inside it, to separate a per-region cost from a per-launch one.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python -c "import torch" 2>/dev/null
for k in 8 20 32 40 64 94; do
  timeout 300 python bench.py --steps $k --warmup 5 --repeats 24 --no-cpu-baseline --no-realtime --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']; t = d['timing']; K = $k
    reg = [a * c * 1e3 / K for a, c in zip(t['regions_ms'], t['kernel_sclk_ghz'])][12:]
    clk = sum(t['kernel_sclk_ghz'][12:]) / len(t['kernel_sclk_ghz'][12:])
    print('K $k', 'value %.4g' % d['value'], 'launches/region %.2f' % (r['kernel_launches'] / 24.0), 'region k-cycles/block (last 12): %.1f' % (sum(reg) / len(reg)),
          'kernel_ms_avg %.4f (all 24 regions; clock of the last 12: %.3f)' % (r['kernel_ms_avg'], clk), 'region ms last:', [round(x, 3) for x in t['regions_ms'][-3:]])
"
done
