#!/bin/bash
# usage: ab_graph.sh <tag> "<variants>" <rounds> <bench args...>
TAG=$1; VARS=$2; ROUNDS=$3; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export OSCEN_GPU_EXPERIMENTAL=1
for r in $(seq 1 $ROUNDS); do for t in $VARS; do
  if [ "$t" = "base" ]; then unset OSCEN_GPU_LIB; else export OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_$t.so; fi
  ( [ -f "$ROOT/oscen_amd/_build/liboscen_gpu_$t.env" ] && set -a && . "$ROOT/oscen_amd/_build/liboscen_gpu_$t.env"; set +a
    timeout 300 python bench.py --no-cpu-baseline --no-realtime --no-configs "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$t', 'round $r', 'value %.4g' % d['value'], 'first5 %.4g' % d['value_median_first5'], 'kernel_ms/block %.5f' % r['kernel_ms_per_block'], r['kernel_variant'])" ) 2>&1 | tee -a $OUT/ab.log
done; done
