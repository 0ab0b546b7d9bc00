"""The `-m gpu` parity tests, executed here -- no GPU -- on the host simulator (tests/hostsim/, TEST INFRASTRUCTURE).

The engine's host code and the very kernel sources the product compiles for gfx950 (csrc/gen/*.hip ahead of time, the
generator's output for every custom graph at run time) are compiled for x86 against a stand-in of <hip/hip_runtime.h>;
a launch runs every lane of a workgroup as a fibre, wave operations and barriers are rendezvous points.  What that checks
on the CPU, through the same C ABI and against the same oracle as on the MI355X: the control logic of the kernels --
chunk variants and sticky loops, the hand-off barriers of the pipelined kernels, event walks, bus tiles, the multirate
schedule -- and the host side of every entry point the selected tests touch.  What it cannot check: anything about the
hardware (numbers agree with the GPU's to the last few bits only: libm instead of ocml, 1/x instead of v_rcp_f32), so the
tests keep the contract's 1e-5 against the oracle and the GPU suite stays the authority.

Nothing of this is reachable from the product: oscen_amd/ never builds, names or loads the simulator; this file builds it
under tests/hostsim/_build/ and points a SUBPROCESS at it through OSCEN_GPU_LIB."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM = os.path.join(ROOT, "tests", "hostsim")


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
    env.pop("OG_HOSTSIM_DEVICES", None)
    return env


def subset():
    with open(os.path.join(HOSTSIM, "subset.txt")) as f:
        return [l.strip() for l in f if l.strip() and not l.startswith("#")]


@pytest.mark.timeout(1500)
def test_gpu_parity_tests_on_the_host_simulator(hostsim_env):
    """~100 of the GPU suite's tests (the ones the simulator finishes in seconds: everything but the full-size banks and
    the latency assertions), each still comparing against the oracle / the interpreter / the reference's known answers"""
    ids = subset()
    assert len(ids) >= 90
    workers = str(max(2, min(8, (os.cpu_count() or 2) // 2)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-n", workers, "--timeout", "300", "-p", "no:cacheprovider"] + ids,
                       cwd=ROOT, env=hostsim_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = r.stdout[-4000:]
    assert r.returncode == 0, tail
    assert "%d passed" % len(ids) in tail, tail


@pytest.mark.timeout(900)
def test_pipelined_kernels_do_not_depend_on_the_order_the_waves_run_in(hostsim_env):
    """Between rendezvous points the simulator may run the ready lanes in any order.  It normally runs wave 0 first; here
    the waves run last-first: the producer wave of a pipelined kernel then reaches every hand-off AFTER its consumers are
    waiting.  A missing or misplaced hand-off barrier (the sticky chunk loops keep theirs inside the loop) changes a result."""
    ids = [t for t in subset() if t.split("::")[0] in ("tests/test_parity_gpu.py", "tests/test_jit_gpu.py", "tests/test_multirate_gpu.py",
                                                       "tests/test_event_edges_gpu.py", "tests/test_voice_grouping_gpu.py")]
    assert len(ids) >= 35
    env = dict(hostsim_env)
    env["OG_HOSTSIM_SCHEDULE"] = "reverse"
    workers = str(max(2, min(8, (os.cpu_count() or 2) // 2)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-n", workers, "--timeout", "300", "-p", "no:cacheprovider"] + ids,
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "%d passed" % len(ids) in r.stdout[-4000:], r.stdout[-4000:]


CLUSTER8 = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import oscen_amd
from tests import oracle_lib as ol

SR = 48000.0
n_dev, per_dev, total, block = 8, 160, 768, 256      # config 4 in small: 8 devices, one shard each, RCCL reduce of the bus
n = n_dev * per_dev
plans = oscen_amd.note_plans(n, span=total)
cl = oscen_amd.Cluster("fm_voice", n, list(range(n_dev)), sample_rate=SR)
assert cl.num_shards == n_dev and cl.num_devices == n_dev, (cl.num_shards, cl.num_devices)
oscen_amd.schedule_note_plans(cl, plans, total_frames=total)
bus = np.concatenate([cl.process_block(block).copy() for _ in range(total // block)], axis=0)
assert cl.rccl_reduces > 0
# one engine over the whole bank: same voices, same note streams
eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
one = eng.render(total, block=block).reshape(total, -1)
mono, abs_sum, _ = ol.render_mt(ol.BANK_FM, 0, n, total, block=block, group=8, seed=oscen_amd.SYNTH_SEED, span=total)
for name, got in (("cluster", bus), ("engine", one)):
    diff = np.abs(got[:, 0].astype(np.float64) - mono)
    assert np.all(diff <= 2e-6 * abs_sum + 1e-5), (name, float((diff / (abs_sum + 1e-30)).max()))
assert np.abs(mono).max() > 0.5
# the two differ only by the association of the cross-shard sum
assert np.max(np.abs(bus - one)) <= 2e-6 * float(abs_sum.max()) + 1e-6
# the batched render (one reduce for all blocks) gives the block-by-block bus bit for bit
cl2 = oscen_amd.Cluster("fm_voice", n, list(range(n_dev)), sample_rate=SR)
oscen_amd.schedule_note_plans(cl2, plans, total_frames=total)
assert np.array_equal(cl2.render(total, block=block).reshape(total, -1), bus)
# every built-in graph, and device lists with several shards per device / in no particular order: the cluster's bus is the
# single engine's (the e-piano's Tremolo runs once, on the root, after the reduce; the echo voice carries delay lines)
for graph, nv in (("epiano_voice", 8 * 24), ("sat4x_voice", 8 * 40), ("echo_voice", 8 * 30)):
    tot, blk = 512, 128
    pl = oscen_amd.note_plans(nv, span=tot)
    single = oscen_amd.Engine(graph, nv, sample_rate=SR)
    oscen_amd.schedule_note_plans(single, pl, total_frames=tot)
    want = single.render(tot, block=blk).reshape(tot, -1)
    for devs in ([0, 1, 2, 3, 4, 5, 6, 7], [0, 0, 1, 1, 2, 2, 3, 3], [3, 1, 2]):
        c = oscen_amd.Cluster(graph, nv, devs, sample_rate=SR)
        assert c.num_devices == len(set(devs))
        oscen_amd.schedule_note_plans(c, pl, total_frames=tot)
        got = np.concatenate([c.process_block(blk).copy() for _ in range(tot // blk)], axis=0).reshape(tot, -1)
        assert got.shape == want.shape and c.rccl_reduces > 0
        assert np.abs(got - want).max() <= 4e-7 * max(1.0, float(np.abs(want).max())) * 4, (graph, devs)
print("cluster8 ok", cl.rccl_reduces, float(np.abs(bus).max()))
"""


@pytest.mark.timeout(600)
def test_eight_device_cluster_with_the_rccl_reduce_on_the_host_simulator(hostsim_env):
    """BASELINE config 4's shape -- one shard per device on EIGHT devices, the stereo-mix reduce over the communicator --
    has never met more than one GPU (no multi-GPU node has been available to any round).  The simulator reports eight
    devices (OG_HOSTSIM_DEVICES) and a stand-in librccl.so.1 sums the ranks' buffers: the cluster's sharding, its
    per-device streams and worker threads, the grouped ncclReduce and the root's hand-over run end to end."""
    env = dict(hostsim_env)
    env["OG_HOSTSIM_DEVICES"] = "8"
    r = subprocess.run([sys.executable, "-c", CLUSTER8 % ROOT], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "cluster8 ok" in r.stdout, r.stdout[-3000:]
