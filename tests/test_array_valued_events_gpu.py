"""Event plumbing inside ARRAY-VALUED voices (several lanes per voice: the electric piano's AmplitudeSource / OscillatorBank
keep `[f32; 32]` fields, one voice spans four lanes).  Round 5: node-to-node event edges and graph event outputs there.

Reference: events are clear + copy per edge (oscen-lib/src/graph/static_context.rs:80-155), a node's handlers run in
process_event_inputs() right before its process() (codegen/emit_node.rs:191-379), an `#[output(event)]` field collects
pushes (oscen-macros/src/lib.rs:237-295).  The model here is the SAME voice gated from the host at the frames the clock
node fires: bit-equal samples, and the graph's event output lists exactly those pushes, once per voice."""
import numpy as np
import pytest

import oscen_amd
from tests.test_event_edges_gpu import CLOCK

pytestmark = pytest.mark.gpu
SR = 48000.0


def ep_graph(clocked, tick_output=True):
    oscen_amd.register_node("Clock::new", CLOCK["inputs"], CLOCK["outputs"], CLOCK["process"], state=CLOCK["state"], n_ctor_args=1,
                            event_outputs=CLOCK["event_outputs"])
    g = oscen_amd.Graph("ep_seq")
    g.input_value("frequency", 220.0, per_voice=True)
    if clocked:
        g.input_value("period", 100.0, per_voice=True)
        g.node("clk", "Clock::new", 100.0)
        g.connect("period", "clk.period")
    else:
        g.input_event("gate")
    g.output_stream("out")
    if clocked and tick_output:
        g.output_event("ticks")
    g.node("amp", "AmplitudeSource::new")
    g.node("bank", "OscillatorBank::new")
    src = "clk.trig" if clocked else "gate"
    g.connect("frequency", "amp.frequency")
    g.connect("frequency", "bank.frequency")
    g.connect(src, "amp.gate")        # electric_piano_voice.rs:373-391: the gate goes to both nodes
    g.connect(src, "bank.gate")
    g.connect("amp.amplitudes", "bank.amplitudes")
    g.connect("bank.output", "out")
    if clocked and tick_output:
        g.connect("clk.trig", "ticks")
    return g


def clock_events(periods, frames):
    """(voice, frame, value) of every push of the Clock nodes: count reaches `period` on frame f when (f + 1) % period == 0"""
    vs, fs, xs = [], [], []
    for v, p in enumerate(periods):
        high = 0
        for f in range(frames):
            if (f + 1) % int(p) == 0:
                high ^= 1
                vs.append(v), fs.append(f), xs.append(0.8 if high else 0.0)
    return np.array(vs), np.array(fs), np.array(xs, dtype=np.float32)


def render(eng, blocks):
    n = eng.n_voices
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    taps, bus = [], []
    for b in blocks:
        bus.append(eng.process_block(b).copy())
        taps.append(eng.read_voice_taps(b))
    return np.concatenate(taps, axis=1), np.concatenate(bus, axis=0)


@pytest.mark.parametrize("n", [16, 17, 70])  # one full wave of 16 voices / a second wave with a single voice / ragged
def test_a_clock_node_gates_an_array_valued_voice_and_feeds_a_graph_event_output(n):
    frames, blocks = 768, (256, 300, 212)
    freqs = np.linspace(80.0, 900.0, n).astype(np.float32)
    periods = (37 + 5 * np.arange(n)).astype(np.float32)
    clk = oscen_amd.Engine(ep_graph(True), n, sample_rate=SR)
    assert clk.lanes_per_voice == 4
    clk.set_voice_values("frequency", freqs)
    clk.set_voice_values("period", periods)
    got, got_bus = render(clk, blocks)
    ev, over = clk.read_output_events()
    host = oscen_amd.Engine(ep_graph(False), n, sample_rate=SR)
    host.set_voice_values("frequency", freqs)
    vs, fs, xs = clock_events(periods, frames)
    host.schedule_voice_events("gate", vs, fs, xs)
    ref, ref_bus = render(host, blocks)
    assert np.abs(ref).max() > 0.5
    assert np.array_equal(got, ref)                       # the in-voice event edge = the same gate from the host, bit for bit
    assert np.allclose(got_bus, ref_bus, rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref_bus).max())))
    # the graph's event output: every push, once per VOICE (not once per lane), with its frame, voice and payload
    assert over == 0 and len(ev) == len(vs)
    want = sorted(zip(fs.tolist(), vs.tolist(), xs.tolist()))
    have = sorted(zip(ev["frame"].tolist(), ev["voice"].tolist(), ev["value"].tolist()))
    assert have == want
    assert clk.events_dropped == 0


def test_block_size_does_not_change_an_array_valued_voice_with_event_edges():
    """block_processing_test.rs:23-286 for this shape: 256 x 2 == 128 x 4 == 512 x 1 == ragged, bit for bit (a lane beyond
    the last voice runs the same tick and its clock fires too: its handler must not write through the state planes)"""
    n = 21
    freqs = np.linspace(90.0, 700.0, n).astype(np.float32)
    periods = (29 + 3 * np.arange(n)).astype(np.float32)
    outs = []
    for blocks in ((256, 256), (128,) * 4, (512,), (100, 1, 411)):
        e = oscen_amd.Engine(ep_graph(True, tick_output=False), n, sample_rate=SR)
        e.set_voice_values("frequency", freqs)
        e.set_voice_values("period", periods)
        outs.append(render(e, blocks)[0])
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    assert np.abs(outs[0]).max() > 0.5


@pytest.mark.parametrize("delay_samples,feedback", [(100.0, 0.0), (37.5, 0.6), (300.25, 0.4)])
def test_delay_line_behind_an_array_valued_voice(delay_samples, feedback):
    """`Delay` (oscen-lib/src/delay/mod.rs:47-83) fed by the electric piano's oscillator bank: the line is one HBM ring per
    VOICE although the voice spans four lanes.  Model: the same voice without the delay (its taps, bit-equal input), then
    the oracle's Delay node sample by sample -- whole-sample and fractional (Catmull-Rom) reads, delays shorter and longer
    than a block."""
    import ctypes as C

    from tests import oracle_lib as ol
    from tests.test_n3_gpu import _delay

    lib = ol.load()
    n, blocks = 37, (128,) * 5
    frames = sum(blocks)
    freqs = np.linspace(80.0, 900.0, n).astype(np.float32)
    vs, fs, xs = clock_events((61 + 7 * np.arange(n)).astype(np.float32), frames)

    def graph(with_delay):
        g = ep_graph(False)
        if with_delay:
            g = oscen_amd.Graph("ep_echo")
            g.input_value("frequency", 220.0, per_voice=True)
            g.input_event("gate")
            g.output_stream("out")
            g.node("amp", "AmplitudeSource::new")
            g.node("bank", "OscillatorBank::new")
            g.node("d", "Delay::new", delay_samples, feedback)
            for s, d in (("frequency", "amp.frequency"), ("frequency", "bank.frequency"), ("gate", "amp.gate"), ("gate", "bank.gate"),
                         ("amp.amplitudes", "bank.amplitudes"), ("bank.output", "d.input"), ("d.output + bank.output", "out")):
                g.connect(s, d)
        return g

    outs = []
    for with_delay in (False, True):
        e = oscen_amd.Engine(graph(with_delay), n, sample_rate=SR)
        assert e.lanes_per_voice == 4
        e.set_voice_values("frequency", freqs)
        e.schedule_voice_events("gate", vs, fs, xs)
        taps, bus = render(e, blocks)
        # the mix bus = the sum of the voices (every voice once, although four lanes hold its output)
        assert np.allclose(bus[:, 0], taps.astype(np.float64).sum(axis=0), rtol=0, atol=1e-5 * max(1.0, float(np.abs(taps).sum(axis=0).max())))
        outs.append(taps)
    dry, wet = outs
    worst = 0.0
    for v in range(n):
        d = _delay(lib, delay_samples, feedback)
        ref = np.zeros(frames, dtype=np.float32)
        for i in range(frames):
            d.input = float(dry[v, i])
            lib.oo_delay_process(C.byref(d))
            ref[i] = np.float32(d.output) + dry[v, i]
        lib.oo_delay_free(C.byref(d))
        worst = max(worst, float(np.max(np.abs(wet[v] - ref) / np.maximum(1.0, np.abs(ref)))))
    assert np.abs(dry).max() > 0.5 and not np.array_equal(dry, wet)
    assert worst <= 1e-5, worst


def test_several_bus_channels_of_an_array_valued_voice():
    """Two stream outputs / a Frame<2> output of a voice whose nodes are array-valued: every channel is summed over the
    VOICES (emit_node.rs:463-466 `voices.out -> out` per output) -- the voice's lead lane contributes, not each of its four
    lanes.  Checked against the one-output graph: channel 0 is its bus, channel 1 the scaled copy; taps likewise."""
    n, blocks = 53, (256, 100)
    freqs = np.linspace(80.0, 900.0, n).astype(np.float32)
    vs, fs, xs = clock_events((61 + 7 * np.arange(n)).astype(np.float32), sum(blocks))

    def graph(kind):
        g = oscen_amd.Graph("ep_" + kind)
        g.input_value("frequency", 220.0, per_voice=True)
        g.input_event("gate")
        if kind == "two":
            g.output_stream("out_a")
            g.output_stream("out_b")
        elif kind == "frame":
            g.output_stream("out")  # (fed a Frame(..): a Frame<2> voice output)
        else:
            g.output_stream("out")
        g.node("amp", "AmplitudeSource::new")
        g.node("bank", "OscillatorBank::new")
        for s, d in (("frequency", "amp.frequency"), ("frequency", "bank.frequency"), ("gate", "amp.gate"), ("gate", "bank.gate"),
                     ("amp.amplitudes", "bank.amplitudes")):
            g.connect(s, d)
        if kind == "two":
            g.connect("bank.output", "out_a")
            g.connect("bank.output * 0.5", "out_b")
        elif kind == "frame":
            g.connect("Frame(bank.output, bank.output * 0.5)", "out")
        else:
            g.connect("bank.output", "out")
        return g

    res = {}
    for kind in ("one", "two", "frame"):
        e = oscen_amd.Engine(graph(kind), n, sample_rate=SR)
        assert e.lanes_per_voice == 4 and e.channels == (1 if kind == "one" else 2)
        e.set_voice_values("frequency", freqs)
        e.schedule_voice_events("gate", vs, fs, xs)
        res[kind] = render(e, blocks)
    taps1, bus1 = res["one"]
    assert np.abs(taps1).max() > 0.5
    for kind in ("two", "frame"):
        taps, bus = res[kind]
        assert taps.shape == taps1.shape + (2,) and bus.shape == (sum(blocks), 2)
        assert np.array_equal(taps[..., 0], taps1) and np.array_equal(taps[..., 1], taps1 * np.float32(0.5))
        scale = max(1.0, float(np.abs(taps1).sum(axis=0).max()))
        assert np.allclose(bus[:, 0], bus1[:, 0], rtol=0, atol=1e-5 * scale)
        assert np.allclose(bus[:, 1], 0.5 * bus1[:, 0], rtol=0, atol=1e-5 * scale)
