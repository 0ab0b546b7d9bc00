#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun).
# usage: scripts/gpu_profile.sh <tag> [extra bench.py args, e.g. --graph echo_voice]   -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (one timed region, no real-time record: the LAST launches of the voice kernel are then exactly the timed ones,
#  which is what scripts/prof_summary.py averages)
CMD="python $ROOT/bench.py --no-cpu-baseline --no-realtime --no-configs --repeats 1 $EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE")
[ -n "${PROF_FULL:-}" ] && PASSES+=("SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES")
for pmc in "${PASSES[@]}"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
echo "$CMD" > $OUT/command.txt
# summarise here (the databases are too large to travel back) and keep only the logs
if [ -n "${PROF_SUMMARY_ARGS:-}" ]; then
  PROF_OUT=$ROOT/gpurun_out/profiles_out python $ROOT/scripts/prof_summary.py $TAG $PROF_SUMMARY_ARGS > $OUT/summary.log 2>&1
  tail -3 $OUT/summary.log
  rm -rf $OUT/stats $OUT/pmc_*/
fi
