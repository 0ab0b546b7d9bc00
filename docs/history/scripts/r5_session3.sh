#!/bin/bash
# Round 5, third GPU session: the whole driver line (configs array, adaptive repeats, shader clock per region), the e-piano
# kernel without its read-mostly tables in registers (suite + profile), full GPU suite.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r05c; mkdir -p $OUT
echo "== (1) the driver's command, whole line"
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_line.json 2> $OUT/driver_line.err ) 2>&1 | grep real
tail -3 $OUT/driver_line.err
python - <<PY
import json
d = json.loads(open('$OUT/driver_line.json').read().strip().splitlines()[-1])
t = d['timing']
print('value %.4g ms_per_step %.4f' % (d['value'], d['ms_per_step']), d['roofline']['kernel_variant'], 'repeats', t['repeats'], 'first5 %.4g first %.4g' % (t['value_median_first5'], t['value_first_region']))
print('regions_ms', ' '.join('%.3f' % x for x in t['regions_ms']))
print('sclk', t['sclk_ghz_after_region'])
for c in d['roofline']['configs']: print({k: c.get(k) for k in ('config','value','ms_per_step','kernel','kernel_ms_per_block','roofline_frac','error')})
rt = d['realtime']
print('realtime', rt['realtime_voices_at_48k'], [(r['voices'], round(r['latency_ms']['p99'],2), r['deadline_misses']) for r in rt['loaded']['runs']], [(r['voices'], round(r['latency_ms']['p99'],2), r['deadline_misses']) for r in rt['idle_bank']['runs']])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
echo "== (2) default run (188 blocks) and the variant line"
for a in "" "--variant survey2" "--voices-per-gpu 1048576"; do
python bench.py --no-cpu-baseline --no-realtime --no-configs $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$a', 'value %.4g' % d['value'], 'kernel_ms/block %.5f' % d['roofline']['kernel_ms_per_block'], d['roofline'].get('kernel_variant'), 'sclk', d['timing']['sclk_ghz_after_region'])
"
done
echo "== (3) the full GPU suite, observed errors"
rm -f $OUT/observed.jsonl
OSCEN_OBSERVED=$OUT/observed.jsonl timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8
python scripts/observed_errors.py $OUT/observed.jsonl > $OUT/observed_errors.md; head -8 $OUT/observed_errors.md; tail -1 $OUT/observed_errors.md
echo "== (4) e-piano profile"
PROF_SUMMARY_ARGS="262144 256 epiano_voice" bash scripts/gpu_profile.sh r05c_epiano --graph epiano_voice --voices-per-gpu 262144 --steps 94
cat $ROOT/gpurun_out/profiles_out/r05c_epiano_summary.md
