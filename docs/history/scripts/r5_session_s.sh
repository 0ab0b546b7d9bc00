#!/bin/bash
# Round 5, session "s": why do 188-block regions cost 59.5 k cycles per block and 94-block regions 69 k?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python -c "import torch" 2>/dev/null
run() { timeout 300 python bench.py "$@" --no-cpu-baseline --no-realtime --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']; t = d['timing']; K = d['steps']
    reg = [a * c * 1e3 / K for a, c in zip(t['regions_ms'], t['kernel_sclk_ghz'])]
    print('$*', '| value %.4g' % d['value'], 'events/region', d['config']['events_in_timed_region'], 'k-cycles/block by region', [round(x, 1) for x in reg])
"; }
run --steps 188 --warmup 8 --repeats 5
run --steps 188 --warmup 5 --repeats 12
run --steps 94 --warmup 8 --repeats 5
run --steps 94 --warmup 8 --repeats 10
run --steps 187 --warmup 8 --repeats 5
run --steps 190 --warmup 8 --repeats 5
run --steps 150 --warmup 8 --repeats 6
run --steps 376 --warmup 8 --repeats 4
run --steps 188 --warmup 8 --repeats 5 --group-voices 0
run --steps 94 --warmup 8 --repeats 10 --group-voices 0
