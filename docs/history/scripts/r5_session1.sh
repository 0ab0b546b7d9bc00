#!/bin/bash
# Round 5, first GPU session: measure what round 4 prepared after its GPU budget ran out.
#   here (no GPU):  python scripts/build_variant.py s1 OGC_STICKY1=1
#   then:           gpurun --timeout 900 -- 'bash scripts/r5_session1.sh > gpurun_out/r5_session1.log 2>&1; tail -60 gpurun_out/r5_session1.log'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT
one() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$1', 'value %.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], d['roofline'].get('kernel_variant'))
"; }
echo "== (1) the new GPU tests of round 4's last commits: voice grouping on the device"
timeout 300 python -m pytest tests/test_voice_grouping_gpu.py tests/test_hard_regime_gpu.py -m gpu -q -x 2>&1 | tail -3
echo "== (2) og_group_voices: driver's command, default run, 131 072 voices (interleaved)"
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | one "driver   plain  "
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --group-voices 2>/dev/null | one "driver   grouped"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --group-voices 2 2>/dev/null | one "driver   policy2"
  python bench.py --no-cpu-baseline --no-realtime 2>/dev/null | one "default  plain  "
  python bench.py --no-cpu-baseline --no-realtime --group-voices 2>/dev/null | one "default  grouped"
done
python bench.py --no-cpu-baseline --no-realtime --voices-per-gpu 131072 2>/dev/null | one "131072   plain  "
python bench.py --no-cpu-baseline --no-realtime --voices-per-gpu 131072 --group-voices 2>/dev/null | one "131072   grouped"
python bench.py --no-cpu-baseline --no-realtime --voices-per-gpu 1048576 2>/dev/null | one "1048576  plain  "
python bench.py --no-cpu-baseline --no-realtime --voices-per-gpu 1048576 --group-voices 2>/dev/null | one "1048576  grouped"
echo "== (2b) grouped bank + release-variant priority (build first: scripts/build_variant.py rp3 OGC_RELPRIO=3)"
if [ -f oscen_amd/_build/liboscen_gpu_rp3.so ]; then
  for r in 1 2 3; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --group-voices 2>/dev/null | one "driver grouped         "
    ( set -a; . oscen_amd/_build/liboscen_gpu_rp3.env; set +a; OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_rp3.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-realtime --group-voices 2>/dev/null | one "driver grouped relprio3" )
  done
fi
echo "== (3) sticky chunks in the ordinary kernel (OGC_STICKY1): 262 144 / 1 M voices, e-piano, saturator"
if [ -f oscen_amd/_build/liboscen_gpu_s1.so ]; then
  bash scripts/ab_bench.sh "base s1" 3 --no-realtime --voices-per-gpu 262144
  bash scripts/ab_bench.sh "base s1" 2 --no-realtime --voices-per-gpu 1048576
  bash scripts/ab_bench.sh "base s1" 2 --no-realtime --graph epiano_voice --voices-per-gpu 262144
  bash scripts/ab_bench.sh "base s1" 2 --no-realtime --graph sat4x_voice --voices-per-gpu 131072
  echo "-- parity of the variant"
  OGC_STICKY1=1 OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_s1.so timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_n3_gpu.py tests/test_epiano_gpu.py tests/test_multirate_gpu.py tests/test_jit_gpu.py -m gpu -q -x -k "not fm262144 and not fm1048576 and not is_jit and not poly_wrapper" 2>&1 | tail -3
else
  echo "(build it first: python scripts/build_variant.py s1 OGC_STICKY1=1)"
fi
echo "== (3b) the hardware sine: accuracy and cost (DESIGN section 8, item 5)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I oscen_amd/csrc -o /tmp/vsin scripts/ubench/vsin.hip 2>/dev/null && /tmp/vsin
echo "== (4) cut and priority sweep of the four-wave kernel now that waves 0-2 are lighter (build the variants first: scripts/build_variant.py c<tag> OGC_CUTS=...)"
# (and the hand-off length: scripts/build_variant.py x4 OGC_XCH=4; scripts/build_variant.py x16 OGC_XCH=16 -- with the sticky
#  loops the static VALU per frame is the same at 4 / 8 / 16 frames per hand-off: 23.3-24.8 / 22.6-24.4 / 22.3-24.2)
TAGS=$(ls oscen_amd/_build/liboscen_gpu_c*.so oscen_amd/_build/liboscen_gpu_p*.so oscen_amd/_build/liboscen_gpu_x*.so 2>/dev/null | sed 's/.*liboscen_gpu_//; s/\.so//' | tr '\n' ' ')
[ -n "$TAGS" ] && bash scripts/ab_bench.sh "base $TAGS" 2 --no-realtime --steps 20 --warmup 5
echo "== (5) the FM sine in turns: st1 = half-turn polynomial (10 VALU), st2 = v_sin_f32 (build: scripts/build_variant.py st1 -- -DOG_SIN_TURNS=1)"
if [ -f oscen_amd/_build/liboscen_gpu_st1.so ]; then
  bash scripts/ab_bench.sh "base st1 st2" 3 --no-realtime --steps 20 --warmup 5
  bash scripts/ab_bench.sh "base st1 st2" 1 --no-realtime
  bash scripts/ab_bench.sh "base st1 st2 s1 st1s1 st2s1" 2 --no-realtime --voices-per-gpu 262144
  for t in base st1 st2; do
    echo "-- parity + observed errors: $t"
    rm -f $OUT/observed_$t.jsonl
    if [ "$t" = "base" ]; then unset OSCEN_GPU_LIB; else export OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_$t.so; fi
    OSCEN_OBSERVED=$OUT/observed_$t.jsonl timeout 500 python -m pytest tests/test_hard_regime_gpu.py tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_midi_gpu.py -m gpu -q -k "not fm1048576 and not is_jit and not poly_wrapper and not largest_real_time" 2>&1 | tail -4
    python scripts/observed_errors.py $OUT/observed_$t.jsonl | head -8
    unset OSCEN_GPU_LIB
  done
fi
