#!/bin/bash
# round 4, session 6: prefetched event records + calibrated depth model: suite, event cost, bench lines, profile of the headline kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r04f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 600 python scripts/dbg_event_cost.py > $OUT/event_cost.log 2>&1; cat $OUT/event_cost.log
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py driver; done
python bench.py --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py default
python bench.py --steps 20 --warmup 5 --sparse-events --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py driver_sparse
for v in 98304 131072; do python bench.py --voices-per-gpu $v --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-realtime 2>/dev/null | python scripts/benchline.py "v$v"; done
PROF_FULL=1 PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh r04f_fm65536 --steps 20 --warmup 5 > $OUT/prof.log 2>&1
tail -3 $OUT/prof.log; sed -n 1,12p gpurun_out/profiles_out/r04f_fm65536_summary.md; grep "SQ_INSTS_VALU\|SQ_WAIT\|SQ_WAVE_CYCLES\|SQ_ACTIVE" gpurun_out/profiles_out/r04f_fm65536_summary.md
