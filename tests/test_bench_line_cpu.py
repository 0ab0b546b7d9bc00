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
torch.cuda.device_count = lambda: int(__import__("os").environ.get("OG_HOSTSIM_DEVICES", "1"))
torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=0)
class _Event:  # (the reduce of a multi-rank run is bracketed by torch events)
    def __init__(self, *a, **k): pass
    def record(self, *a, **k): pass
    def elapsed_time(self, other): return 0.0
torch.cuda.Event = _Event
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
    d = strict_line(line)
    detail = [ln for ln in r.stdout.splitlines() if ln.startswith("BENCH_DETAIL ")]
    assert len(detail) == 1 and r.stdout.index(detail[0]) < r.stdout.index(line)  # the full record comes BEFORE the line
    d["_detail"] = json.loads(detail[0][len("BENCH_DETAIL "):])
    return d


def _no_constants(name):
    raise ValueError("non-strict JSON constant %s in the bench line" % name)


def strict_line(line):
    """The line as the driver reads it: one line, < 8000 bytes (a log's 8 KB tail holds it whole), strict JSON (no NaN /
    Infinity), the contract's keys."""
    assert "\n" not in line and len(line.encode()) < 8000, len(line)
    d = json.loads(line, parse_constant=_no_constants)
    for k in CONTRACT:
        assert k in d, k
    return d


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
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_hash", "stale_profile", "dram_gbs", "kernel_variant",
              "kernel_ms_per_block", "traffic_source", "valu_issue"):
        assert k in rf, k
    assert "value_median_first5" in d and d["detail"] and os.path.exists(os.path.join(ROOT, d["detail"]))
    assert "regions_ms" not in d["timing"] and len(d["_detail"]["timing"]["regions_ms"]) == 2  # every region: the detail record
    names = [c[0] for c in d["configs"]]
    assert names == ["2-variant", "4-shard", "3a", "3b", "5"]
    for c in d["configs"]:
        assert c[-1] is None, c                                   # no error
        assert c[2] > 0 and c[3] > 0 and isinstance(c[4], str) and c[4].startswith("og_k")
    full = d["_detail"]["configs_detail"]
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


@pytest.mark.timeout(900)
def test_eight_rank_line_carries_config4_and_the_og_cluster_leg(hostsim_env, tmp_path):
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, one rank per device; gloo here) on the
    simulator's eight devices: the line has the rank-per-GPU headline, the `config4` sub-record (262 144 voices per GPU,
    scaled down here) through the same ranks, and the `og_cluster` sub-record -- the product's own multi-GPU leg
    (og_cluster_*, the library's ncclReduce over a stand-in librccl) over all eight devices, asserted to span them."""
    script = tmp_path / "bench_harness.py"
    argv = ["--gpus", "8", "--steps", "4", "--warmup", "2", "--repeats", "2", "--voices-per-gpu", "64", "--backend", "gloo", "--test-scale", "2048"]
    script.write_text(HARNESS % {"root": ROOT, "argv": argv, "bench": os.path.join(ROOT, "bench.py")})
    env = dict(hostsim_env)
    env["OG_HOSTSIM_DEVICES"] = "8"
    env["OMP_NUM_THREADS"] = "1"
    # torch has the real librccl.so.1 in the process already: bind the simulator's stand-in by path
    env["OSCEN_GPU_RCCL_LIB"] = os.path.join(os.path.dirname(env["OSCEN_GPU_LIB"]), "fake_rccl", "librccl.so.1")
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                         # rank 0 prints ONE line
    d = strict_line(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["config"]["total_voices"] == 8 * 64 and d["scaling"] == "weak"
    assert d["multi_gpu"]["rccl_ranks"] == 8 and len(d["multi_gpu"]["per_rank_kernel_ms_avg"]) == 8
    c4 = d["config4"]
    assert c4["voices_per_gpu"] == 128 and c4["total_voices"] == 8 * 128 and c4["rccl_ranks"] == 8 and c4["value"] > 0
    full = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("BENCH_DETAIL ")][0][len("BENCH_DETAIL "):])
    assert len(full["config4"]["multi_gpu"]["per_rank_kernel_ms_avg"]) == 8
    oc = d["og_cluster"]
    assert "error" not in oc, oc
    assert oc["rccl_ranks"] == 8 and oc["cluster"]["devices"] == 8 and oc["cluster"]["rccl_reduces"] > 0 and oc["value"] > 0
    # round 6: the run's wall clock leg by leg (so that an 8-GPU run can be seen to fit a timeout), and the og_cluster leg
    # started only after the other seven ranks had released their devices (a file each; rank 0 waits for all of them)
    assert set(d["wall_s"]) >= {"headline", "config4", "wait_for_ranks_to_release_devices", "og_cluster", "total_since_start"}, d["wall_s"]
    assert oc.get("ranks_released_devices_first") is True, oc
    assert "configs" not in d and d["cpu_baseline"] is None       # N = 1 extras stay at N = 1


def test_pipeline_depth_by_bank_size(hostsim_env):
    """The engine's choice of kernel shape (og_engine.cpp, depth model; profiles/r05e_depth_sweep.md): the four-wave pipeline
    for small banks, TWO waves at 131 072 voices (eight workgroups per CU, of which only six four-wave ones are resident: a
    short second round costs almost a whole one), four again at 196 608 (two full rounds), the ordinary kernel from 262 144
    on.  The simulator reports the occupancies the round-5 fm kernels have on gfx950 (16 / 8 / 6 workgroups per CU)."""
    code = ("import sys; sys.path.insert(0, %r)\nimport oscen_amd\n"
            "for V in (64, 16384, 65536, 98304, 131072, 196608, 262144, 1048576):\n"
            "    e = oscen_amd.Engine('fm_voice', V, sample_rate=48000.0); print(V, e.pipeline_depth); e.close()\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=hostsim_env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = dict(tuple(int(x) for x in ln.split()) for ln in r.stdout.strip().splitlines())
    assert got == {64: 4, 16384: 4, 65536: 4, 98304: 4, 131072: 2, 196608: 4, 262144: 1, 1048576: 1}, got


def test_the_line_survives_a_leg_that_never_returns(tmp_path):
    """bench.py's watchdog (N > 1: the og_cluster sub-record is the one leg no rank-per-GPU phase has exercised before it
    runs): a leg that hangs -- here a sleep standing in for a collective that never completes -- costs its own sub-record,
    not the line; a leg that returns in time cancels the watchdog."""
    code = """
import sys, time, json
sys.path.insert(0, %r)
import bench
line = {"metric": "m", "value": 1.0}
g = bench.Watchdog(60, lambda: bench.emit_line_and_exit(line, {"og_cluster": {"error": "fast leg timed out"}}))
g.cancel()                                   # a leg that came back
g = bench.Watchdog(1, lambda: bench.emit_line_and_exit(line, {"og_cluster": {"error": "timed out"}, "cpu_baseline": None}))
print("noise before the line")
time.sleep(30)                               # a leg that does not
print(json.dumps({"never": "printed"}))
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=25)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    d = json.loads(last)
    assert d["metric"] == "m" and d["value"] == 1.0 and d["og_cluster"] == {"error": "timed out"} and d["cpu_baseline"] is None
    assert "never" not in d


def test_compact_line_of_a_full_driver_record_fits_a_log_tail():
    """Round 5's driver record (profiles/bench_r05q_driver.json: 20 KB, every leg present -- 45 regions, six real-time banks, the
    CPU scaling probe, five configurations) through compact_line(): under 8000 bytes, strict JSON, the keys VERDICT r5 item 1
    lists.  A NaN anywhere in the record becomes null instead of a bare NaN token."""
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r05q_driver.json")))
    full["realtime"]["idle_bank"]["runs"] = full["realtime"]["runs"]
    full["roofline"]["dram_frac"] = float("nan")
    full["timing"]["value_max"] = float("inf")
    line = bench.compact_line(full)
    d = strict_line(line)
    assert d["timing"]["value_max"] is None
    assert d["value"] == full["value"] and d["value_median_first5"] and d["realtime_voices_at_48k"] == 8388608
    assert len(d["realtime"]["runs"]) == 6 and d["realtime"]["runs"][-1]["loaded"] and d["realtime"]["deadline_ms"]
    assert [c[0] for c in d["configs"]] == ["2-variant", "4-shard", "3a", "3b", "5"]
    for k in ("value", "unit", "cores", "kind", "sample", "single_thread"):
        assert k in d["cpu_baseline"], k
    for k in ("frac", "achieved", "peak", "dram_gbs", "kernel_variant", "kernel_ms_per_block", "traffic_source", "stale_profile"):
        assert k in d["roofline"], k
    assert d["roofline"]["valu_issue"]["frac"] > 0
