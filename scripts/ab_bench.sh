#!/bin/bash
# interleaved A/B of library variants: scripts/ab_bench.sh "<tag> <tag> ..." [rounds] [bench args]
TAGS=${1:-"base"}; ROUNDS=${2:-3}; shift; shift
for r in $(seq 1 $ROUNDS); do
  for t in $TAGS; do
    unset OGC_TPT_LAZY OGC_RELPRIO OGC_XCH OGC_STICKY1 OGC_STICKY OGC_SLOWPRIO OGC_RELPRIO OGC_ROT OGC_PRIO OGC_WAVES_EU OSCEN_GPU_SPLIT OGC_CUTS OGC_EVUNROLL OGC_ALAP OGC_FCNT OGC_TPT_CHUNK OGC_HPL OGC_WAVES_EU OGC_EVSKIP OGC_EVUNROLL OGC_FORCE_PATH OGC_UNROLL OGC_ROT OGC_PARTS OGC_CUT2 OGC_NOSYNC OGC_PRIO OGC_PRIO_PARITY OGC_K3 OGC_TPT_FLAT OGC_CHUNK_CHK OGC_SPLIT
    export OSCEN_GPU_EXPERIMENTAL=1  # the OGC_* knobs are ignored without it (og_abi.h)
    # a variant's build environment must also be in force at run time (the engine re-derives the kernel hash)
    if [ -f "$PWD/oscen_amd/_build/liboscen_gpu_$t.env" ]; then set -a; . "$PWD/oscen_amd/_build/liboscen_gpu_$t.env"; set +a; fi
    if [ "$t" = "base" ]; then unset OSCEN_GPU_LIB; else export OSCEN_GPU_LIB=$PWD/oscen_amd/_build/liboscen_gpu_$t.so; fi
    case $t in u[0-9]*) export OGC_UNROLL=${t#u};; esac
    python bench.py --steps 94 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$t', 'round $r', 'value %.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'], 'ms_per_step %.4f' % d['ms_per_step'])
"
  done
done
