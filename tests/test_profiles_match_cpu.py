"""The committed rocprofv3 summaries (profiles/*_summary.json) are of the kernels THIS tree builds: `bench.py` looks its
PMC figures (`roofline.traffic`, `valu_issue`) up by kernel hash and reports `stale_profile` when the newest summary of a
configuration was taken on another build -- VERDICT r4 asked for `stale_profile: null` in the driver's line.  This test
fails the moment a change to the generator or the device headers renames the kernels without the profiles being re-taken
(docs/history/scripts/r6_final.sh <tag> prof)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oscen_amd  # noqa: E402


def kernel_hash(graph):
    return re.search(r"\bog_k_([0-9a-f]{16})_00\b", oscen_amd.Graph(builtin=graph).kernel_source()).group(1)


def test_every_configuration_bench_py_prints_has_a_profile_of_the_current_kernels():
    cases = [("fm_voice", 65536, None, 20.0), ("fm_voice", 65536, None, 188 / 6.0), ("fm_voice", 1048576, None, 188 / 24.0)]
    for _, graph, voices, steps, variant in bench.OTHER_CONFIGS:
        cases.append((graph, voices, variant, None))
    hashes = {}
    for graph, voices, variant, bpl in cases:
        h = hashes.setdefault(graph, kernel_hash(graph))
        p = bench.pmc_profile(voices, 256, graph, h, bpl, variant)
        assert p["source"] is not None and p["stale"] is None, (graph, voices, variant, h, p)
        assert p["bytes"] and p["valu"], (graph, voices, variant)
