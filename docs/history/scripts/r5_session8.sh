#!/bin/bash
# why is the watched-input TPT test slower where the cutoff moves every frame?  instruction counts of both builds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for t in base tl0; do
  if [ "$t" = "base" ]; then unset OSCEN_GPU_LIB OGC_TPT_LAZY; else export OSCEN_GPU_LIB=$ROOT/oscen_amd/_build/liboscen_gpu_tl0.so OGC_TPT_LAZY=0; fi
  rm -rf /tmp/pmc_$t
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS -d /tmp/pmc_$t -o pmc -- python $ROOT/bench.py --no-cpu-baseline --no-realtime --no-configs --repeats 1 --variant survey2 > /tmp/pmc_$t.log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmc_$t/**/pmc_results.db', recursive=True)
c = sqlite3.connect(db[0])
rows = {}
for name, kern, value in c.execute("select counter_name, kernel_name, value from counters_collection where kernel_name like 'og_k%' order by dispatch_id"):
    rows.setdefault((name, kern[-2:]), []).append(value)
for k, v in sorted(rows.items()):
    print('$t', k, 'dispatches', len(v), 'last-7 avg %.4g' % (sum(v[-7:]) / len(v[-7:])))
PY
done
