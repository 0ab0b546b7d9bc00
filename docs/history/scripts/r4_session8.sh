#!/bin/bash
# round 4, session 8: slow-path priority adopted -- fm profiles again, driver line, suite, fm lines, K=2 check
TAG=r04p
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536 --steps 20 --warmup 5
PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_default
PROF_SUMMARY_ARGS="262144 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm262144 --voices-per-gpu 262144
cd $ROOT
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
bash scripts/r4_final.sh $TAG tests
timeout 600 python bench.py --no-realtime > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --sparse-events --no-cpu-baseline --no-realtime > $OUT/bench_driver_sparse.json 2> /dev/null
for v in 262144 1048576; do timeout 300 python bench.py --voices-per-gpu $v --no-cpu-baseline --no-realtime > $OUT/bench_fm_$v.json 2> /dev/null; done
# forced two-wave kernel: is the slow-path priority also right there?
( for r in 1 2; do for t in base nosp; do
    if [ $t = base ]; then unset OSCEN_GPU_LIB OGC_SLOWPRIO; else export OSCEN_GPU_LIB=$PWD/oscen_amd/_build/liboscen_gpu_nosp.so OGC_SLOWPRIO=-1; fi
    OSCEN_GPU_SPLIT=2 python bench.py --steps 94 --warmup 4 --no-cpu-baseline --no-realtime 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$t', 'k2 value %.4g' % d['value'], d['roofline'].get('kernel_variant'))
"; done; done ) > $OUT/ab_k2b.log 2>&1
unset OSCEN_GPU_LIB OGC_SLOWPRIO
cat $OUT/ab_k2b.log
timeout 300 python bench.py --midi-live 1000 --no-cpu-baseline --no-realtime > $OUT/bench_midi_live.json 2> /dev/null
timeout 300 python bench.py --dist-single --steps 20 --warmup 5 --no-cpu-baseline --no-realtime > $OUT/bench_rccl1.json 2> /dev/null
timeout 300 python bench.py --graph sub_voice --voices-per-gpu 262144 --steps 94 --no-cpu-baseline --no-realtime > $OUT/bench_sub.json 2> /dev/null
