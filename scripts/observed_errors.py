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
