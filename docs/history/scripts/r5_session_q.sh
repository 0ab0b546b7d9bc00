#!/bin/bash
# Round 5, session "q": the blocks of a timed region handed over in one call (og_process_blocks_async) against one call per
# block, interleaved; the new entry's tests; the driver's command once more for the record.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r05q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_bench_entry_points_gpu.py "tests/test_fullsize_gpu.py::test_the_timed_path_equals_the_checked_path_bit_for_bit" -m gpu -q 2>&1 | tail -3
show() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    t = d['timing']; r = d['roofline']
    print('$1', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'first5 %.4g' % t['value_median_first5'], 'kern_ms/block %.5f' % r['kernel_ms_per_block'], d['config'].get('host_calls','')[:40], 'last regions', [round(x,3) for x in t['regions_ms'][-4:]])
"; }
for r in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --no-configs 2>/dev/null | show batched
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --no-configs --per-block-calls 2>/dev/null | show per-block
done
timeout 300 python bench.py --no-cpu-baseline --no-realtime --no-configs 2>/dev/null | show default-batched
timeout 300 python bench.py --no-cpu-baseline --no-realtime --no-configs --per-block-calls 2>/dev/null | show default-per-block
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
cat $OUT/bench_driver.json | show driver
