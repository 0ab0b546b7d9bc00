#!/bin/bash
# Round 5, fourth GPU session: everything since session 3 on the hardware -- v_fract phase wrap, array-valued event edges /
# Delay / bus channels, frame_offset on in-voice events, ADVICE fixes, the kernel's own clock reading.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r05d; mkdir -p $OUT
echo "== (1) the full GPU suite, observed errors"
rm -f $OUT/observed.jsonl
OSCEN_OBSERVED=$OUT/observed.jsonl timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -12
python scripts/observed_errors.py $OUT/observed.jsonl > $OUT/observed_errors.md; head -8 $OUT/observed_errors.md; tail -1 $OUT/observed_errors.md
echo "== (2) the driver's command (no real-time record), default, 262 144, 1 M, variant: value + the kernel's own clock"
for a in "--steps 20 --warmup 5" "" "--voices-per-gpu 262144" "--voices-per-gpu 1048576" "--variant survey2" "--steps 20 --warmup 5 --repeats 5"; do
python bench.py --no-cpu-baseline --no-realtime --no-configs $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    t = d['timing']
    print('[$a]', 'value %.4g' % d['value'], 'kernel_ms/block %.5f' % d['roofline']['kernel_ms_per_block'], d['roofline'].get('kernel_variant'))
    print('   regions_ms', ' '.join('%.3f' % x for x in t['regions_ms'][:48]))
    print('   kernel sclk', t['kernel_sclk_ghz'][:48])
    print('   probe  sclk', t['sclk_ghz_after_region'][:48])
"
done
echo "== (3) the other graphs"
for a in "--graph epiano_voice --voices-per-gpu 262144 --steps 94" "--graph sub_voice --voices-per-gpu 262144 --steps 94" "--graph sat4x_voice --voices-per-gpu 131072 --steps 94"; do
python bench.py --no-cpu-baseline --no-realtime --no-configs $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('[$a]', 'value %.4g' % d['value'], 'kernel_ms/block %.5f' % d['roofline']['kernel_ms_per_block'], d['roofline'].get('kernel_variant'), 'kclk', d['timing']['kernel_sclk_ghz'])
"
done
