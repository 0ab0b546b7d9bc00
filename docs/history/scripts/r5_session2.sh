#!/bin/bash
# Round 5, second GPU session: the new defaults (v_sin_f32 operators, sticky chunks in the ordinary kernel) -- full GPU suite
# with the observed-error record, wave placement, stage rotation A/B, clock ramp over the regions, the configs array.
#   gpurun --timeout 1500 -- 'bash scripts/r5_session2.sh > gpurun_out/r5_session2.log 2>&1; tail -5 gpurun_out/r5_session2.log'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT
echo "== (1) where the waves of the four-wave kernel land"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid scripts/ubench/hwid.hip 2>/dev/null && /tmp/hwid
echo "== (2) stage rotation of the four-wave kernel (OGC_ROT): driver's command, interleaved"
bash scripts/ab_bench.sh "base r0 r2 r3 r4" 3 --no-realtime --no-configs --steps 20 --warmup 5
echo "== (3) clock ramp: the same 20-step region 48 times (regions_ms), then 188-step regions"
python bench.py --steps 20 --warmup 5 --repeats 48 --no-cpu-baseline --no-realtime --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('value(median) %.4g' % d['value'], 'first %.4g' % d['timing']['value_first_region'], 'max %.4g' % d['timing']['value_max'])
    print('regions_ms', ' '.join('%.3f' % x for x in d['timing']['regions_ms']))
"
echo "== (4) the driver's command, whole line (configs array, real-time record, CPU baseline): wall time"
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_line.json 2> $OUT/driver_line.err; grep -E "Elapsed|Maximum resident" $OUT/driver_line.err
python -c "
import json
d = json.loads(open('$OUT/driver_line.json').read().strip().splitlines()[-1])
print('value %.4g ms_per_step %.4f' % (d['value'], d['ms_per_step']), d['roofline']['kernel_variant'], 'stale', d['roofline']['stale_profile'])
print(json.dumps(d['configs']))
print('realtime', d['realtime']['realtime_voices_at_48k'], [ (r['voices'], round(r['latency_ms']['p99'],2), r['deadline_misses']) for r in d['realtime']['loaded']['runs']])
"
echo "== (5) the full GPU suite on the new defaults, observed errors"
rm -f $OUT/observed.jsonl
OSCEN_OBSERVED=$OUT/observed.jsonl timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6
python scripts/observed_errors.py $OUT/observed.jsonl > $OUT/observed_errors.md; head -12 $OUT/observed_errors.md; tail -2 $OUT/observed_errors.md
