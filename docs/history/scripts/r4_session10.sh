#!/bin/bash
# round 4, session 10: sticky chunks adopted -- fm profiles and every fm bench line again
TAG=r04t
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536 --steps 20 --warmup 5
PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_default
PROF_SUMMARY_ARGS="262144 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm262144 --voices-per-gpu 262144
cd $ROOT
# the new summaries are what bench.py matches by kernel hash: put them in place before the bench lines
cp gpurun_out/profiles_out/${TAG}_*_summary.* profiles/
( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
timeout 300 python bench.py --no-realtime > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 100 python bench.py --steps 20 --warmup 5 --sparse-events --no-cpu-baseline --no-realtime > $OUT/bench_driver_sparse.json 2> /dev/null
for v in 262144 1048576; do timeout 100 python bench.py --voices-per-gpu $v --no-cpu-baseline --no-realtime > $OUT/bench_fm_$v.json 2> /dev/null; done
timeout 100 python bench.py --midi-live 1000 --no-cpu-baseline --no-realtime > $OUT/bench_midi_live.json 2> /dev/null
timeout 100 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-realtime > $OUT/bench_rccl1.json 2> /dev/null
timeout 100 python bench.py --graph sub_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime > $OUT/bench_sub.json 2> /dev/null
