#!/bin/bash
# rocprofv3 --kernel-trace --stats of the BLOCKING real-time entry (scripts/dbg_rt_hiccup.py: og_midi_send_batch of 1000
# messages + og_midi_process_block(256), one launch per block) at a given bank size -> profiles-style markdown.
# usage (on the GPU box, through gpurun): scripts/prof_realtime.sh <tag> <voices> <blocks> [loaded]
#   loaded: with the synthetic score resident on every voice (the loaded bank of bench.py's real-time record)
set -u
TAG=${1:-r03g_rt8m}; V=${2:-8388608}; N=${3:-400}; MODE=${4:-}
EXTRA=""; WHAT="og_midi_send_batch of 1 000 live MIDI messages"
if [ "$MODE" = "loaded" ]; then EXTRA="back loaded"; WHAT="the synthetic score resident on every voice + og_midi_send_batch of 1 000 live MIDI messages"; fi
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $ROOT/gpurun_out/profiles_out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/scripts/dbg_rt_hiccup.py $V $N $EXTRA > $OUT/stats.log 2>&1
python - <<PY
import sqlite3, os
con = sqlite3.connect(os.path.join("$OUT", "stats", "stats_results.db"))
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"))
md = ["# rocprofv3 summary \`$TAG\`: the blocking real-time entry", "",
      "\`rocprofv3 --kernel-trace --stats -- python scripts/dbg_rt_hiccup.py $V $N $EXTRA\` ($WHAT + "
      "og_midi_process_block(256) per block, one launch per block, $V voices, MI355X)", "",
      "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
for name, calls, total, avg, pct in rows:
    md.append("| \`%s\` | %d | %.1f | %.1f |" % (name[:70], calls, avg / 1e3 if avg > 1e5 else avg, pct))
per_block = sum(r[2] for r in rows) / max(1, $N + 50)
md += ["", "GPU time per block (all kernels, %d blocks incl. 50 warm-up): %.1f us of the 5 333 us block period; the host-side wall latency "
       "of the same loop is in stats.log / bench.py's real-time record." % ($N + 50, per_block / 1e3 if per_block > 1e5 else per_block)]
open(os.path.join("$ROOT", "gpurun_out", "profiles_out", "${TAG}_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
PY
tail -3 $OUT/stats.log
rm -rf $OUT/stats
