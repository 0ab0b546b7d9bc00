#!/bin/bash
# Round 5, session "o": interleaved A/B of the shipped library (base) against the r05n build (bus column instead of a
# per-frame mask; wave-uniform lazy TPT test) on every bench configuration.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out; mkdir -p $OUT
run() { # tag args...
  local t=$1; shift
  if [ "$t" = "base" ]; then unset OSCEN_GPU_LIB; else export OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_$t.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-realtime --no-configs "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$t', '$*', 'value %.4g' % d['value'], 'kern_ms/block %.5f' % d['roofline']['kernel_ms_per_block'], d['roofline']['kernel_variant'])
"
}
python -c "import torch" 2>/dev/null
for r in 1 2 3; do
  for t in base r05n; do
    run $t --steps 20 --warmup 5
    run $t
    run $t --variant survey2
    run $t --voices-per-gpu 262144
    run $t --graph sub_voice --voices-per-gpu 262144 --steps 94
    [ $r -lt 3 ] && run $t --graph epiano_voice --voices-per-gpu 262144 --steps 94
    [ $r -lt 2 ] && run $t --graph sat4x_voice --voices-per-gpu 131072 --steps 94
  done
done
