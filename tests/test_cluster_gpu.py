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
    monkeypatch.setenv("OSCEN_GPU_EXPERIMENTAL", "1")  # (OSCEN_GPU_FORCE_RCCL is an experiment knob: ignored without it)
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


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4 through the product path, the moment the box has more than one GPU: a cluster over REAL devices
# (one engine per GPU, RCCL communicator with one rank per device, one ncclReduce per launch batch) at config 4's
# per-GPU shard size.  Skips cleanly on a one-GPU box; needs nothing but more devices to run.
# ---------------------------------------------------------------------------------------------------------------
def device_count():
    import ctypes as C

    oscen_amd.load_library()
    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")  # the runtime liboscen_gpu.so is linked against
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


@pytest.mark.parametrize("graph", ["fm_voice", "epiano_voice"])
def test_multi_device_cluster_at_the_config4_shard_size(graph):
    n_dev = device_count()
    if n_dev < 2:
        pytest.skip("needs at least two GPUs (the driver's multi-GPU node)")
    from tests import oracle_lib as ol
    from tests.test_fullsize_gpu import oracle_taps, sample_voices

    kind = ol.BANK_FM if graph == "fm_voice" else ol.BANK_EPIANO
    devs = list(range(min(n_dev, 8)))
    per_gpu = 262144
    n, total, block = per_gpu * len(devs), 512, 256
    cl = oscen_amd.Cluster(graph, n, devs, sample_rate=SR)
    assert cl.num_shards == len(devs) and cl.num_devices == len(devs)
    plans = oscen_amd.note_plans(n, span=total)
    oscen_amd.schedule_note_plans(cl, plans, total_frames=total)
    # taps: sampled GLOBAL voices, set on the shard that owns each
    voices = sample_voices(n, 96)
    shards = [cl.shard(s) for s in range(cl.num_shards)]
    owned = []
    for sh in shards:
        mine = voices[(voices >= sh.first_voice) & (voices < sh.first_voice + sh.n_voices)]
        sh.set_voice_taps((mine - sh.first_voice).astype(np.uint32))
        owned.append(mine)
    bus, taps = [], [[] for _ in shards]
    for _ in range(total // block):
        bus.append(cl.process_block(block).copy())
        for k, sh in enumerate(shards):
            if len(owned[k]):
                taps[k].append(sh.read_voice_taps(block))
    bus = np.concatenate(bus, axis=0)
    assert cl.rccl_reduces > 0  # the RCCL leg really ran: one reduce per launch batch over the devices' communicator
    # (1) sampled voices of every shard against the oracle
    got = np.concatenate([np.concatenate(t, axis=1) for t in taps if t], axis=0)
    ref = oracle_taps(kind, np.concatenate([o for o in owned if len(o)]), total, block)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert float(err.max()) <= 1e-5, float(err.max())
    # (2) the reduced bus against the oracle's f64 sum over ALL voices (the e-piano through the Tremolo pan sequence,
    #     which runs once on the root after the reduce)
    mono, abs_sum, _ = ol.render_mt(kind, 0, n, total, block=block, group=8, seed=oscen_amd.SYNTH_SEED, span=total)
    if graph == "epiano_voice":
        pan = ol.tremolo_pan(total, 5.0, 0.3, SR).astype(np.float64)
        want, scale = mono[:, None] * pan, abs_sum[:, None] * np.ones((1, 2))
    else:
        want, scale = mono[:, None], abs_sum[:, None]
    assert bus.shape == want.shape
    diff = np.abs(bus.astype(np.float64) - want)
    assert np.all(diff <= 2e-6 * scale + 1e-5), float((diff / (scale + 1e-30)).max())
    assert np.abs(want).max() > 1.0
    # (3) the batched render (one ncclReduce for all blocks) gives the same bus as block by block
    cl2 = oscen_amd.Cluster(graph, n, devs, sample_rate=SR)
    oscen_amd.schedule_note_plans(cl2, plans, total_frames=total)
    bus2 = cl2.render(total, block=block)
    assert cl2.rccl_reduces == 1
    assert np.array_equal(bus2, bus)


def test_stereo_voice_outputs_through_the_cluster(monkeypatch):
    """a graph whose stream output is fed a Frame<2> (per-voice pan): the shards hand over interleaved L R sums, the
    reduce covers both channels"""
    monkeypatch.setenv("OSCEN_GPU_EXPERIMENTAL", "1")  # (OSCEN_GPU_FORCE_RCCL is an experiment knob: ignored without it)
    monkeypatch.setenv("OSCEN_GPU_FORCE_RCCL", "1")
    n, total, block = 1500, 768, 256
    oscen_amd.register_node(
        "ClPan::new", inputs=[("input", "stream", 0.0, -1), ("pan", "value", 0.5, -1)], outputs=[("output", 2)],
        process="    output.v[0] = input * (1.0f - pan);\n    output.v[1] = input * pan;\n")
    try:
        g = oscen_amd.Graph("cl_panned")
        g.input_value("frequency", 220.0, per_voice=True)
        g.input_value("pan", 0.5, per_voice=True)
        g.input_event("gate")
        g.output_stream("out")
        g.node("osc", "PolyBlepOscillator::saw", 220.0, 0.25)
        g.node("env", "AdsrEnvelope::new", 0.005, 0.05, 0.7, 0.05)
        g.node("p", "ClPan::new")
        g.connect("frequency", "osc.frequency")
        g.connect("gate", "env.gate")
        g.connect("pan", "p.pan")
        g.connect("osc.output * env.output", "p.input")
        g.connect("p.output", "out")
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        cl = oscen_amd.Cluster(g, n, [0, 0, 0], sample_rate=SR)
        assert cl.channels == 2 and eng.channels == 2
        f = np.linspace(110.0, 1760.0, n).astype(np.float32)
        pan = np.linspace(0.0, 1.0, n).astype(np.float32)
        for x in (eng, cl):
            x.set_voice_values("frequency", f)
            x.set_voice_values("pan", pan)
            x.schedule_voice_events("gate", np.arange(n), 3 + np.arange(n) % 50, np.full(n, 0.8, np.float32))
        want = eng.render(total, block=block)
        got = cl.render(total, block=block)
        assert cl.rccl_reduces == 1
        assert got.shape == want.shape == (total, 2)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.max(np.abs(got - want)) <= 1e-4 * scale
        assert np.abs(want[:, 0]).max() > 1e-2 and np.abs(want[:, 0] - want[:, 1]).max() > 1e-3
        per_block = oscen_amd.Cluster(g, n, [0, 0], sample_rate=SR)
        per_block.set_voice_values("frequency", f)
        per_block.set_voice_values("pan", pan)
        per_block.schedule_voice_events("gate", np.arange(n), 3 + np.arange(n) % 50, np.full(n, 0.8, np.float32))
        pb = np.concatenate([per_block.process_block(block) for _ in range(total // block)], axis=0)
        assert np.max(np.abs(pb - want)) <= 1e-4 * scale
    finally:
        oscen_amd.unregister_node("ClPan::new")


def test_eight_shard_cluster_on_one_device_matches_the_single_engine(monkeypatch):
    """the shape of BASELINE config 4 (eight shards, one reduce per batch) at a small size with all shards on the one
    device there is: global voice ranges, per-shard note streams, the per-device accumulation of eight buffers, the RCCL
    leg with a one-rank communicator; render (threaded, batched) and the per-block real-time entry (thread-free)"""
    monkeypatch.setenv("OSCEN_GPU_EXPERIMENTAL", "1")  # (OSCEN_GPU_FORCE_RCCL is an experiment knob: ignored without it)
    monkeypatch.setenv("OSCEN_GPU_FORCE_RCCL", "1")
    n, total, block = 8 * 640 + 37, 1280, 256   # (a ragged total: the shards differ in size)
    want = single("fm_voice", n, total, block)
    got, cl = cluster("fm_voice", n, total, block, shards=8)
    assert cl.num_shards == 8 and cl.num_devices == 1 and cl.rccl_reduces == 1
    sizes = [cl.shard(s).n_voices for s in range(8)]
    assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    scale = max(1.0, float(np.abs(want).max()))
    assert np.max(np.abs(got - want)) <= 1e-4 * scale and np.abs(want).max() > 1e-2
    per_block, cl2 = cluster("fm_voice", n, total, block, shards=8, per_block=True)
    assert np.array_equal(per_block, got)           # the two entries agree bit for bit
    assert cl2.rccl_reduces == total // block       # one reduce per block through the real-time entry


def test_graph_event_outputs_over_a_cluster_equal_the_single_engine():
    """og_cluster_read_output_events: every shard logs the events of its own voices; merged they are the single engine's
    list -- (frame, GLOBAL voice, push order) -- and the dropped pushes add up (og_cluster_events_dropped)"""
    oscen_amd.register_node(
        "ClBurst::new", inputs=[("period", "value", 100.0, 0), ("pushes", "value", 1.0, 1)], outputs=["level"], n_ctor_args=2,
        state=[("count", "u32", 0, -1), ("fired", "f32", 0.0, -1)], event_outputs=["tick"],
        process="""
    count += 1u;
    if ((float)count >= period) {
        count = 0u;
        fired += 1.0f;
        for (int k = 0; k < (int)pushes; ++k) tick.push(fired * 10.0f + (float)k);
    }
    level = fired;
""")
    try:
        g = oscen_amd.Graph("cl_bursts")
        g.input_value("period", 100.0, per_voice=True)
        g.input_value("pushes", 1.0, per_voice=True)
        g.output_stream("out")
        g.output_event("ticks")
        g.node("b", "ClBurst::new", 100.0, 1.0)
        g.connect("period", "b.period")
        g.connect("pushes", "b.pushes")
        g.connect("b.level", "out")
        g.connect("b.tick", "ticks")
        n = 333
        periods = (37 + (np.arange(n) * 7) % 90).astype(np.float32)
        pushes = (1 + np.arange(n) % 3).astype(np.float32)  # a third push on one frame is dropped and counted
        one = oscen_amd.Engine(g, n, sample_rate=SR)
        cl = oscen_amd.Cluster(g, n, [0, 0, 0], sample_rate=SR)
        for x in (one, cl):
            x.set_voice_values("period", periods)
            x.set_voice_values("pushes", pushes)
        for frames in (256, 100, 412):
            a = one.process_block(frames)
            b = cl.process_block(frames)
            assert np.allclose(a, b, rtol=1e-6, atol=1e-5)
        ev1, over1 = one.read_output_events()
        ev2, over2 = cl.read_output_events()
        assert over1 == 0 and over2 == 0 and len(ev1) > 1000
        assert np.array_equal(ev1, ev2)
        assert ev2["voice"].max() > 300  # global voice ids
        assert one.events_dropped == cl.events_dropped > 0
        ev3, _ = cl.read_output_events()
        assert len(ev3) == 0
    finally:
        oscen_amd.unregister_node("ClBurst::new")
