#!/bin/bash
# PMC comparison of the fm_voice kernel with and without events in the timed region (scripts/dbg_event_cost.py W0 / W6)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/evpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|SQ_IFETCH|SQ_WAIT_INST|INST_LEVEL|SQC_" | head -60 > $OUT/avail.txt
for w in W0 W6; do
  for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
    name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
    timeout 200 rocprofv3 --pmc $pmc -d $OUT/${w}_$name -o pmc -- python $ROOT/scripts/dbg_event_cost.py $w > $OUT/${w}_$name.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/evpmc"
for w in ("W0", "W6"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/%s_*/**/*counter_collection.csv" % w, recursive=True):
        for r in csv.DictReader(open(f)):
            if "og_k" not in r.get("Kernel_Name", ""): continue
            a = acc[(r["Kernel_Name"][:24], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(w, k, c, "sum %.4g launches %d per-launch %.4g" % (v, n, v / max(n, 1)))
PY
