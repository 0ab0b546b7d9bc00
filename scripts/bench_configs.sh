#!/bin/bash
# the BASELINE.json configs on one GPU (values for BASELINE.md / DESIGN.md); run through gpurun
run() { python bench.py --steps 94 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    r = d['roofline']
    print('$*', '| value %.4g' % d['value'], '| kernel_ms %.4f' % r['kernel_ms_avg'], '| ms/step %.4f' % d['ms_per_step'], '| B/vs %.3f' % r['bytes_per_voice_sample'], '| GB/s %.1f' % r['achieved'], '| rt voices %.3g' % d['realtime_voices_at_48k'])
"; }
run --graph fm_voice --voices-per-gpu 65536
run --graph fm_voice --voices-per-gpu 131072
run --graph fm_voice --voices-per-gpu 262144
run --graph fm_voice --voices-per-gpu 1048576
run --graph sub_voice --voices-per-gpu 262144
run --graph epiano_voice --voices-per-gpu 262144
run --graph sat4x_voice --voices-per-gpu 131072
run --graph sat1x_voice --voices-per-gpu 131072
run --graph echo_voice --voices-per-gpu 65536
run --graph echo_voice --voices-per-gpu 262144
