"""Row (e) behind the C ABI: og_cluster_* (contiguous global voice shards, one batched bus reduce).  On the
one-GPU test box the shards share a device: the per-device accumulation runs for real, and
OSCEN_GPU_FORCE_RCCL=1 also drives the RCCL leg (dlopen, ncclCommInitAll, one ncclReduce per batch) with a
one-rank communicator.  The reference semantics: the bus is the sum over ALL voices
(oscen-graph-compiler/src/codegen/emit_node.rs:463-466); a post-mix Tremolo runs after that sum
(examples/electric-piano/src/main.rs:88-96)."""
import os

import numpy as np
import pytest

import oscen_amd

pytestmark = pytest.mark.gpu
SR = 48000.0


def single(graph, n, total, block):
    eng = oscen_amd.Engine(graph, n, sample_rate=SR)
    plans = oscen_amd.note_plans(n, span=total)
    oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
    return eng.render(total, block=block)


def cluster(graph, n, total, block, shards, per_block=False):
    cl = oscen_amd.Cluster(graph, n, [0] * shards, sample_rate=SR)
    plans = oscen_amd.note_plans(n, span=total)
    cl.set_voice_values("frequency", plans["frequency"])
    try:
        gi = cl.index("gate")
        oscen_amd.schedule_note_plans(cl, plans, total_frames=total)
    except oscen_amd.OscenError:
        gi = None
    if per_block:
        out = np.concatenate([cl.process_block(block) for _ in range(total // block)], axis=0)
    else:
        out = cl.render(total, block=block)
    return out, cl


@pytest.mark.parametrize("graph,n", [("fm_voice", 3000), ("epiano_voice", 300), ("sat4x_voice", 1000)])
def test_two_shard_cluster_matches_the_single_engine_bus(graph, n, monkeypatch):
    monkeypatch.setenv("OSCEN_GPU_FORCE_RCCL", "1")
    total, block = 1024, 256
    want = single(graph, n, total, block)
    got, cl = cluster(graph, n, total, block, shards=2)
    assert cl.num_shards == 2 and cl.num_devices == 1 and cl.rccl_reduces == 1
    assert got.shape == want.shape == (total, 2 if graph == "epiano_voice" else 1)
    # same voices, same kernels; only the association of the bus sum differs (two partial sums instead of one tree)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.max(np.abs(got - want)) <= 2e-6 * n * scale / 50 + 1e-5 * scale
    assert np.abs(want).max() > 1e-2


def test_cluster_block_by_block_equals_cluster_render_and_shard_counts_agree():
    total, block, n = 1536, 256, 5000
    a, _ = cluster("fm_voice", n, total, block, shards=3)
    b, _ = cluster("fm_voice", n, total, block, shards=3, per_block=True)
    assert np.array_equal(a, b)  # batching the reduce does not change a bit
    c, _ = cluster("fm_voice", n, total, block, shards=1)
    want = single("fm_voice", n, total, block)
    assert np.array_equal(c, want)  # one shard = the engine itself
    assert np.max(np.abs(a - want)) <= 1e-4 * max(1.0, float(np.abs(want).max()))


def test_cluster_routes_per_voice_calls_by_global_voice_id():
    n = 1000
    cl = oscen_amd.Cluster("fm_voice", n, [0, 0, 0], sample_rate=SR)
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    f = np.linspace(110.0, 880.0, n).astype(np.float32)
    cl.set_voice_values("frequency", f)
    eng.set_voice_values("frequency", f)
    for v in (0, 332, 333, 334, 666, 999):  # around the shard boundaries 333 / 666
        assert cl.push_voice_event("gate", v, v % 200, 0.9) == 0
        assert eng.push_voice_event("gate", v, v % 200, 0.9) == 0
    cl.set_value("filter_cutoff", 3000.0)
    eng.set_value("filter_cutoff", 3000.0)
    for _ in range(3):
        a = cl.process_block(256)
        b = eng.process_block(256)
        assert np.max(np.abs(a - b)) <= 1e-5 * max(1.0, float(np.abs(b).max()))
    assert np.abs(b).max() > 1e-2
    assert cl.push_voice_event("gate", n, 0, 1.0) == oscen_amd.OG_E_INVALID
