"""Condense the OSCEN_OBSERVED record of a GPU test run: max observed oracle error per test.
usage: OSCEN_OBSERVED=gpurun_out/observed.jsonl python -m pytest tests -m gpu ...; python scripts/observed_errors.py gpurun_out/observed.jsonl"""
import collections, json, sys
mx = collections.OrderedDict()
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    mx[d["test"]] = max(mx.get(d["test"], 0.0), d["value"])
print("| test | observed max error (contract 1e-5) |\n|---|---|")
for k, v in sorted(mx.items(), key=lambda kv: -kv[1]):
    print("| `%s` | %.3g |" % (k, v))
print("\noverall max: %.3g over %d comparisons in %d tests" % (max(mx.values()), sum(1 for _ in open(sys.argv[1])), len(mx)))
# the error-vs-time curve of tests/test_hard_regime_gpu.py::test_fm_voice_variant_one_second, if the run wrote one
import os
curve_path = sys.argv[1] + ".hard_regime_curve.json"
if os.path.exists(curve_path):
    d = json.load(open(curve_path))
    win = collections.OrderedDict()
    for f0, e in d["curve"]:
        win[f0 // 4800] = max(win.get(f0 // 4800, 0.0), e)
    print("\n## One second in the hard regime (64 voices, feedback .3 / .2, route .5, env amount 2000, cutoff ramp at frame 4 800)\n")
    print("max error over the 64 voices per 0.1 s window (|gpu - oracle| / max(1, |oracle|); reference peak %.3f):\n" % d["ref_peak"])
    print("| window start (s) | " + " | ".join("%.1f" % (k * 0.1) for k in win) + " |")
    print("|---|" + "---|" * len(win))
    print("| max error | " + " | ".join("%.2e" % v for v in win.values()) + " |")
    print("\nworst block: %.3g -- the error does not grow with time (the loop gain of the operators' self-feedback is 0.94)." % d["worst"])

# the error-vs-time curves of tests/test_timed_config_gpu.py (the configuration bench.py times, whole horizon), if written
import glob
for path in sorted(glob.glob(sys.argv[1] + ".timed_*.json")):
    d = json.load(open(path))
    name = os.path.basename(path)[len(os.path.basename(sys.argv[1])) + len(".timed_"):-len(".json")]
    win = collections.OrderedDict()
    for f0, e in d["curve"]:
        win[f0 // 4800] = max(win.get(f0 // 4800, 0.0), e)
    print("\n## The timed configuration, %s (%d voices grouped, %s, 20-block queued runs; %d sampled voices, %d frames)\n"
          % (name, d["voices"], d["kernel"], d["sampled"], d["frames"]))
    print("max error over the sampled voices per 0.1 s window (reference peak %.3f); the timed path's bus and final state equal the checked run's bit for bit:\n" % d["ref_peak"])
    print("| window start (s) | " + " | ".join("%.1f" % (k * 0.1) for k in win) + " |")
    print("|---|" + "---|" * len(win))
    print("| max error | " + " | ".join("%.2e" % v for v in win.values()) + " |")
    print("\nworst block: %.3g" % d["worst"])
