"""Frame<N> stream payloads inside a voice (oscen-lib/src/frame.rs, graph/static_context.rs:41-69,
182-194, filters/tpt/mod.rs:104-123): user nodes with Frame<2> ports, copy and element-wise fan-in sum of frame edges,
`frame * f32` and `frame - frame` compound sources, TptFilter::<Frame<2>> with one integrator pair per channel.
Checked against a per-sample model over the oracle's nodes (whose TptFilter restatement carries the two channels)."""
import ctypes as C

import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol
from tests.graph_interp import _make

pytestmark = pytest.mark.gpu
SR = 48000.0
f32 = np.float32

DSL = """
name: stereo;
input frequency: value = 220.0;
input cutoff: value = 1500.0;
input gate: event;
output out: stream;
nodes {
    osc = PolyBlepOscillator::saw(220.0, 1.0);
    env = AdsrEnvelope::new(0.01, 0.05, 0.6, 0.02);
    a = Spread::new(0.3);
    b = Spread::new(0.8);
    filt = TptFilter::<Frame<2>>::new(1500.0, 0.9);
    mix = Downmix::new();
}
connections {
    frequency -> osc.frequency;
    gate -> env.gate;
    cutoff -> filt.cutoff;
    osc.output -> a.input;
    osc.output * env.output -> b.input;
    a.output * env.output + (b.output - a.output) -> filt.input;
    filt.output -> mix.input;
    mix.output -> out;
}
"""


def register():
    oscen_amd.register_node(
        "Spread::new", inputs=[("input", "stream", 0.0, -1), ("pan", "value", 0.5, 0)], outputs=[("output", 2)], n_ctor_args=1,
        process="    output.v[0] = input * (1.0f - pan);\n    output.v[1] = input * pan;\n")
    oscen_amd.register_node(
        "Downmix::new", inputs=[("input", "stream", 0.0, -1, 2)], outputs=["output"],
        process="    output = input.v[0] - 0.5f * input.v[1];\n")


def model(freqs, cutoffs, gates, frames):
    lib = ol.load()
    ref = np.zeros((len(freqs), frames), dtype=np.float32)
    for v in range(len(freqs)):
        osc = _make(lib, f32(SR), "PolyBlepOscillator::saw", [220.0, 1.0])
        env = _make(lib, f32(SR), "AdsrEnvelope::new", [0.01, 0.05, 0.6, 0.02])
        osc.set("frequency", freqs[v])
        filt = ol.Tpt()
        lib.oo_tpt_new(C.byref(filt), 1500.0, 0.9, 2)
        filt.sample_rate = SR
        lib.oo_tpt_prepare(C.byref(filt))
        filt.cutoff = float(cutoffs[v])
        for f in range(frames):
            osc.process()
            for gf, gv in gates[v]:
                if gf == f:
                    env.gate(gv)
            env.process()
            o, e = osc.get("output"), env.get("output")
            pa, pb = f32(0.3), f32(0.8)
            a = (f32(o * f32(f32(1.0) - pa)), f32(o * pa))
            ob = f32(o * e)
            b = (f32(ob * f32(f32(1.0) - pb)), f32(ob * pb))
            for ch in range(2):  # connect, then accumulate in edge order
                filt.input[ch] = float(f32(f32(a[ch] * e) + f32(b[ch] - a[ch])))
            lib.oo_tpt_process(C.byref(filt))
            ref[v, f] = f32(f32(filt.output[0]) - f32(f32(0.5) * f32(filt.output[1])))
    return ref


@pytest.mark.parametrize("split", [0, 2])
def test_frame_edges_and_the_stereo_filter(split, monkeypatch):
    register()
    monkeypatch.setenv("OSCEN_GPU_SPLIT", str(split))  # the ordinary kernel, and the channels crossing a pipeline cut
    g = oscen_amd.Graph(dsl=DSL, per_voice=("frequency", "cutoff"))
    n, frames = 80, 700
    freqs = np.linspace(60.0, 3000.0, n).astype(np.float32)
    cutoffs = np.linspace(300.0, 9000.0, n).astype(np.float32)
    gates = [[(5 + v, 0.9), (300 + 2 * v, 0.0), (500 + v, 0.7)] for v in range(n)]
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_values("cutoff", cutoffs)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    for v in range(n):
        for gf, gv in gates[v]:
            eng.schedule_voice_event("gate", v, gf, gv)
    got = []
    for b in (256, 188, 256):
        eng.process_block(b)
        got.append(eng.read_voice_taps(b))
    got = np.concatenate(got, axis=1)
    assert eng.pipeline_depth == max(1, split)
    ref = model(freqs, cutoffs, gates, frames)
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5 and float(np.abs(ref).max()) > 0.05, err


def test_stereo_filter_channels_are_independent():
    """filters/tpt/mod.rs `stereo_channels_are_independent`: an impulse on channel 0 reproduces the mono impulse response
    (the reference's own known answers) on channel 0 and leaves channel 1 silent"""
    oscen_amd.register_node(
        "ImpulseL::new", inputs=[], outputs=[("output", 2)], state=[("fired", "u32", 0, -1)],
        process="    output.v[0] = fired ? 0.0f : 1.0f;\n    output.v[1] = 0.0f;\n    fired = 1u;\n")
    oscen_amd.register_node(
        "Pick::new", inputs=[("input", "stream", 0.0, -1, 2), ("which", "value", 0.0, 0)], outputs=["output"], n_ctor_args=1,
        process="    output = which > 0.5f ? input.v[1] : input.v[0];\n")
    expected = np.array([0.014401104, 0.052318562, 0.089890145, 0.11065749, 0.11862421, 0.11729243, 0.10961619, 0.098000914],
                        dtype=np.float32)
    for which, want in ((0.0, expected), (1.0, np.zeros(8, dtype=np.float32))):
        g = oscen_amd.Graph("imp")
        g.output_stream("out")
        g.node("i", "ImpulseL::new")
        g.node("f", "TptFilter::<Stereo>::new", 2000.0, 0.707)
        g.node("p", "Pick::new", which)
        g.connect("i.output", "f.input")
        g.connect("f.output", "p.input")
        g.connect("p.output", "out")
        eng = oscen_amd.Engine(g, 1, sample_rate=SR)
        out = eng.process_block(8)[:, 0]
        assert np.allclose(out, want, atol=1e-6), (which, out)


def test_frame2_voice_output_gives_a_stereo_bus():
    """the graph's stream output fed a Frame<2> (a per-voice pan): both channels are summed over the voices, the bus is
    interleaved L R, taps are [tap][frame][2]; also across the multi-pass bus reduce (> 1024 workgroups) and with
    several blocks per launch"""
    oscen_amd.register_node(
        "Pan::new", inputs=[("input", "stream", 0.0, -1), ("pan", "value", 0.5, -1)], outputs=[("output", 2)],
        process="    output.v[0] = input * (1.0f - pan);\n    output.v[1] = input * pan;\n")
    g = oscen_amd.Graph("panned")
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_value("pan", 0.5, per_voice=True)
    g.input_event("gate")
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 220.0, 0.25)
    g.node("env", "AdsrEnvelope::new", 0.005, 0.05, 0.7, 0.05)
    g.node("p", "Pan::new")
    g.connect("frequency", "osc.frequency")
    g.connect("gate", "env.gate")
    g.connect("pan", "p.pan")
    g.connect("osc.output * env.output", "p.input")
    g.connect("p.output", "out")
    lib = ol.load()
    for n, blocks in ((200, (256, 300, 212)), (70000, (256, 256, 256))):
        freqs = (55.0 * (1.0 + np.arange(n) % 97)).astype(np.float32)
        pans = ((np.arange(n) * 37 % 101) / 100.0).astype(np.float32)
        probe = np.unique(np.linspace(0, n - 1, 40).astype(np.uint32))
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        assert eng.channels == 2 and eng.lib.og_voice_channels(eng.h) == 2  # (any kernel shape: the pipelined ones carry several bus channels since round 4)
        eng.set_voice_values("frequency", freqs)
        eng.set_voice_values("pan", pans)
        eng.schedule_voice_events("gate", np.arange(n), 3 + np.arange(n) % 50, np.full(n, 0.8, np.float32))
        eng.set_voice_taps(probe)
        bus, taps = [], []
        for b in blocks:
            bus.append(eng.process_block(b))
            taps.append(eng.read_voice_taps(b))
        bus, taps = np.concatenate(bus), np.concatenate(taps, axis=1)
        frames = sum(blocks)
        assert bus.shape == (frames, 2) and taps.shape == (len(probe), frames, 2)
        # per-voice model over the oracle's nodes (all voices for the small bank, the probed ones for the big one)
        voices = range(n) if n <= 1000 else [int(v) for v in probe]
        ref = {}
        for v in voices:
            osc = _make(lib, f32(SR), "PolyBlepOscillator::saw", [220.0, 0.25])
            env = _make(lib, f32(SR), "AdsrEnvelope::new", [0.005, 0.05, 0.7, 0.05])
            osc.set("frequency", freqs[v])
            y = np.zeros((frames, 2), dtype=np.float32)
            for f in range(frames):
                osc.process()
                if f == 3 + v % 50:
                    env.gate(0.8)
                env.process()
                x = f32(osc.get("output") * env.get("output"))
                y[f, 0], y[f, 1] = f32(x * f32(f32(1.0) - pans[v])), f32(x * pans[v])
            ref[v] = y
        for i, v in enumerate(probe):
            r = ref[int(v)]
            assert float(np.max(np.abs(taps[i] - r) / np.maximum(1.0, np.abs(r)))) <= 1e-5
        if n <= 1000:
            want = np.sum([ref[v].astype(np.float64) for v in voices], axis=0)
            scale = np.sum([np.abs(ref[v]).astype(np.float64) for v in voices], axis=0)
            assert np.all(np.abs(bus - want) <= 2e-6 * np.maximum(scale, 1e-3)) and np.abs(want).max() > 1.0
        # several blocks per launch give the same bits as a launch per block
        e2 = oscen_amd.Engine(g, n, sample_rate=SR)
        e2.set_voice_values("frequency", freqs)
        e2.set_voice_values("pan", pans)
        e2.schedule_voice_events("gate", np.arange(n), 3 + np.arange(n) % 50, np.full(n, 0.8, np.float32))
        assert np.array_equal(e2.render(frames, block=256)[: blocks[0]], bus[: blocks[0]])
