"""bench.py end to end on the host simulator (tests/hostsim/, TEST INFRASTRUCTURE): the JSON line of the contract -- the
headline record, `roofline`, the `configs` array of the other BASELINE configurations and the `--variant survey2` line --
is produced by the real script over the real engine code, with torch's device calls redirected to host memory (the
simulator's device memory IS host memory).  Numbers mean nothing here; keys, shapes and the control flow do."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM = os.path.join(ROOT, "tests", "hostsim")

HARNESS = r"""
import sys, types
sys.path.insert(0, %(root)r)
import torch
# redirect torch's device calls to the host: the simulator's "device" pointers are host pointers
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda: None
torch.cuda.device_count = lambda: 1
torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=0)
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{**k, "device": "cpu"})
sys.argv = ["bench.py"] + %(argv)r
import runpy
runpy.run_path(%(bench)r, run_name="__main__")
"""


@pytest.fixture(scope="module")
def hostsim_env():
    sys.path.insert(0, HOSTSIM)
    try:
        import build_hostsim
    finally:
        sys.path.pop(0)
    lib = build_hostsim.build()
    env = dict(os.environ)
    env["OSCEN_GPU_LIB"] = lib
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(lib), "fake_rccl") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    return env


def run_bench(env, argv):
    code = HARNESS % {"root": ROOT, "argv": argv, "bench": os.path.join(ROOT, "bench.py")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert r.stdout.rstrip().endswith(line)  # the JSON is the LAST line of stdout
    return json.loads(line)


CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"]


@pytest.mark.timeout(900)
def test_default_line_carries_every_baseline_configuration(hostsim_env):
    d = run_bench(hostsim_env, ["--steps", "4", "--warmup", "2", "--repeats", "2", "--voices-per-gpu", "192", "--no-realtime", "--no-cpu-baseline",
                                "--test-scale", "2048"])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 192 * 4 * 256 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_hash", "stale_profile"):
        assert k in rf, k
    assert list(d)[-2:] == ["configs", "configs_keys"]            # last keys of the line: they survive a log's tail
    names = [c[0] for c in d["configs"]]
    assert names == ["2-variant", "4-shard", "3a", "3b", "5"]
    for c in d["configs"]:
        assert c[-1] is None, c                                   # no error
        assert c[2] > 0 and c[3] > 0 and isinstance(c[4], str) and c[4].startswith("og_k")
    full = rf["configs"]
    assert [c["graph"] for c in full] == ["fm_voice", "fm_voice", "epiano_voice", "sub_voice", "sat4x_voice"]
    assert full[0]["variant"] == "survey2" and full[0]["kernel"].endswith(("_10", "_11")) or "kernel" in full[0]
    assert full[2]["events_in_timed_region"] >= 0 and full[4]["events_in_timed_region"] == 0  # the saturator has no gate
    assert len(json.dumps(d["configs"])) + len(json.dumps(d["configs_keys"])) < 1900


@pytest.mark.timeout(600)
def test_variant_line(hostsim_env):
    d = run_bench(hostsim_env, ["--steps", "24", "--warmup", "2", "--repeats", "2", "--voices-per-gpu", "128", "--no-realtime", "--no-cpu-baseline",
                                "--variant", "survey2"])
    assert d["config"]["variant"] == "survey2" and "configs" not in d
    assert d["value"] > 0
