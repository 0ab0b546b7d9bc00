#!/bin/bash
# round 4, session 2: depth-first schedule order + cut variants, A/B; the suite; the loaded real-time record
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04b; mkdir -p $OUT
bash scripts/ab_bench.sh "base cut5 cut6 kahn strict" 3 --no-realtime --steps 20 --warmup 5 > $OUT/ab_driver.log 2>&1
bash scripts/ab_bench.sh "base cut5 cut6 kahn" 2 --no-realtime > $OUT/ab_default.log 2>&1
for s in 4 0; do OSCEN_GPU_SPLIT=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py split$s; done > $OUT/split.log 2>&1
cat $OUT/ab_driver.log $OUT/ab_default.log $OUT/split.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04b/bench_driver.json").read().strip().splitlines()[-1])
print("value %.4g" % d["value"], "rt", d.get("realtime_voices_at_48k"))
rt=d["realtime"]
for r in rt["loaded"]["runs"]:
    print("loaded", r["voices"], r["blocks"], r["latency_ms"], "miss", r["deadline_misses"], "setup_s %.1f first_ms %.1f" % (r["setup_s"], r["first_block_ms"]), r["event_stats"], r.get("note"), r.get("paced"))
for r in rt["idle_bank"]["runs"]:
    print("idle", r["voices"], r["latency_ms"]["p50"], r["latency_ms"]["p99"], r["latency_ms"]["max"], r["deadline_misses"])
PY
tail -3 $OUT/bench_driver.err
