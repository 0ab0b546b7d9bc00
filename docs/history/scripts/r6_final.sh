#!/bin/bash
# Round 6 measurement session (run after the LAST kernel change): the GPU suite with the observed-error record, the driver's
# command, counter profiles of every configuration bench.py prints.
# usage: scripts/r6_final.sh <tag> [parts: tests bench prof]      -> gpurun_out/<tag>/, gpurun_out/profiles_out/
TAG=${1:-r06f}; shift || true
PARTS=${*:-"tests bench prof"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for p in $PARTS; do case $p in
tests)
  rm -f $OUT/observed.jsonl*
  OSCEN_OBSERVED=$OUT/observed.jsonl timeout 1700 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  grep -E "passed|failed|FAILED|rc=" $OUT/pytest.log | tail -8
  python scripts/observed_errors.py $OUT/observed.jsonl > $OUT/observed_errors.md 2>&1; head -6 $OUT/observed_errors.md; grep "overall max" $OUT/observed_errors.md ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log ;;
bench)
  ( time OSCEN_BENCH_DETAIL=$OUT/bench_driver_detail.json timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err ) 2>&1 | grep real
  tail -1 $OUT/bench_driver.out > $OUT/bench_driver.json
  python - $OUT/bench_driver.json <<'PY'
import json, sys
l = open(sys.argv[1]).read()
d = json.loads(l); r = d["roofline"]; v = r.get("valu_issue") or {}
print("line bytes", len(l), "value %.4g first5 %.4g ms/step %.4f kern_ms/block %.5f frac %.4f dram %s valu %s %s stale %s" % (
    d["value"], d["value_median_first5"], d["ms_per_step"], r["kernel_ms_per_block"], r["frac"], r.get("dram_gbs"), v.get("frac"), r.get("kernel_variant"), r.get("stale_profile")))
for c in d.get("configs") or []: print("   cfg", c)
print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "rt", d.get("realtime_voices_at_48k"))
for r2 in (d.get("realtime") or {}).get("runs", []): print("   rt", r2)
PY
  ;;
bench_more)
  for spec in "default:" "fm1048576:--voices-per-gpu 1048576" "survey2:--variant survey2" "midi_live:--midi-live 1000" "rccl1:--dist-single --steps 20 --warmup 5" "echo:--graph echo_voice --voices-per-gpu 262144 --steps 94" "k188:--steps 188 --warmup 8"; do
    n=${spec%%:*}; a=${spec#*:}
    OSCEN_BENCH_DETAIL=$OUT/bench_${n}_detail.json timeout 600 python bench.py --no-cpu-baseline --no-realtime --no-configs $a 2> /dev/null | tail -1 > $OUT/bench_$n.json
    python -c "
import json,sys
d=json.load(open('$OUT/bench_$n.json')); r=d['roofline']
print('$n', 'value %.4g ms/step %.4f kern_ms/block %.5f %s' % (d['value'], d['ms_per_step'], r['kernel_ms_per_block'], r['kernel_variant']))" ;
  done ;;
prof)
  PROF_FULL=1 PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536 --steps 20 --warmup 5
  PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_default
  PROF_SUMMARY_ARGS="65536 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm65536_survey2 --variant survey2
  PROF_SUMMARY_ARGS="262144 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm262144 --voices-per-gpu 262144
  PROF_SUMMARY_ARGS="1048576 256 fm_voice" bash scripts/gpu_profile.sh ${TAG}_fm1048576 --voices-per-gpu 1048576
  PROF_SUMMARY_ARGS="262144 256 epiano_voice" bash scripts/gpu_profile.sh ${TAG}_epiano --graph epiano_voice --voices-per-gpu 262144 --steps 94
  PROF_SUMMARY_ARGS="262144 256 sub_voice" bash scripts/gpu_profile.sh ${TAG}_sub --graph sub_voice --voices-per-gpu 262144 --steps 94
  PROF_SUMMARY_ARGS="131072 256 sat4x_voice" bash scripts/gpu_profile.sh ${TAG}_sat4x --graph sat4x_voice --voices-per-gpu 131072 --steps 94
  ls $ROOT/gpurun_out/profiles_out/ ;;
esac; done
