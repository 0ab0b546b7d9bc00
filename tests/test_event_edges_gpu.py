"""Node-to-node event edges inside a voice ("events = clear + copy, last write wins";
oscen-lib/src/graph/static_context.rs:84-155, oscen-macros/src/lib.rs:237-285, event_passthrough.rs): a user node
with an `#[output(event)]` field clocks an AdsrEnvelope, a handler forwards events, EventPassthrough is routing only,
and on fan-in only the last connected source delivers.  Checked against per-sample models over the oracle's nodes."""
import numpy as np
import pytest

import oscen_amd
from tests import oracle_lib as ol
from tests.graph_interp import _make

pytestmark = pytest.mark.gpu
SR = 48000.0
f32 = np.float32

CLOCK = dict(
    inputs=[("period", "value", 100.0, 0), ("velocity", "value", 0.8, -1)], outputs=[], n_ctor_args=1,
    state=[("count", "u32", 0, -1), ("high", "u32", 0, -1)], event_outputs=["trig"],
    process="""
    count += 1u;
    if ((float)count >= period) { count = 0u; high ^= 1u; trig.push(high ? velocity : 0.0f); }
""")
# forwards gate events at half velocity (a handler that pushes), and counts them in a stream output
HALVER = dict(
    inputs=[("input", "event", 0.0, -1)], outputs=["seen"], state=[("n", "f32", 0.0, -1)], event_outputs=["output"],
    process="    seen = n;\n", handlers={"input": "    n += 1.0f; output.push(value * 0.5f);\n"})


def clocked_voice(extra=None):
    oscen_amd.register_node("Clock::new", CLOCK["inputs"], CLOCK["outputs"], CLOCK["process"], state=CLOCK["state"],
                            n_ctor_args=1, event_outputs=CLOCK["event_outputs"])
    g = oscen_amd.Graph("seq")
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_value("period", 100.0, per_voice=True)
    g.output_stream("out")
    g.node("clk", "Clock::new", 100.0)
    g.node("env", "AdsrEnvelope::new", 0.002, 0.01, 0.6, 0.01)
    g.node("osc", "PolyBlepOscillator::saw", 220.0, 1.0)
    g.connect("period", "clk.period")
    g.connect("frequency", "osc.frequency")
    if extra:
        extra(g)
    else:
        g.connect("clk.trig", "env.gate")
    g.connect("osc.output * env.output", "out")
    return g


def model(freqs, periods, frames, gate_scale=1.0):
    lib = ol.load()
    ref = np.zeros((len(freqs), frames), dtype=np.float32)
    for v in range(len(freqs)):
        env = _make(lib, f32(SR), "AdsrEnvelope::new", [0.002, 0.01, 0.6, 0.01])
        osc = _make(lib, f32(SR), "PolyBlepOscillator::saw", [220.0, 1.0])
        osc.set("frequency", freqs[v])
        count, high = 0, 0
        for f in range(frames):
            count += 1
            trig = None
            if f32(count) >= f32(periods[v]):
                count, high = 0, high ^ 1
                trig = f32(0.8) if high else f32(0.0)
            osc.process()
            if trig is not None:
                env.gate(f32(trig * f32(gate_scale)))
            env.process()
            ref[v, f] = f32(osc.get("output") * env.get("output"))
    return ref


def render(g, freqs, periods, frames, blocks=(256, 300, 212)):
    n = len(freqs)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_values("period", periods)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    got = []
    for b in blocks:
        eng.process_block(b)
        got.append(eng.read_voice_taps(b))
    assert eng.pipeline_depth == 1  # the ordinary kernel: node-to-node events do not cross pipeline waves
    return np.concatenate(got, axis=1)


def close(got, ref):
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    assert err <= 1e-5 and float(np.abs(ref).max()) > 0.1, err


def test_a_clock_node_gates_the_envelope_of_its_voice():
    n, frames = 96, 768
    freqs = np.linspace(80.0, 2500.0, n).astype(np.float32)
    periods = (37 + 5 * np.arange(n)).astype(np.float32)  # every voice its own tempo: stage ends and gates on any frame
    got = render(clocked_voice(), freqs, periods, frames)
    close(got, model(freqs, periods, frames))


def test_handler_that_pushes_and_passthrough_chain():
    """clk.trig -> half.input (on_input pushes value/2) -> pass1 -> pass2 -> env.gate"""
    oscen_amd.register_node("Halver::new", HALVER["inputs"], HALVER["outputs"], HALVER["process"], state=HALVER["state"],
                            handlers=HALVER["handlers"], event_outputs=HALVER["event_outputs"])

    def wire(g):
        g.node("half", "Halver::new")
        g.node("pass1", "EventPassthrough::new")
        g.node("pass2", "EventPassthrough::new")
        g.connect("clk.trig", "half.input")
        g.connect("half.output", "pass1.input")
        g.connect("pass1.output", "pass2.input")
        g.connect("pass2.output", "env.gate")

    n, frames = 70, 768
    freqs = np.linspace(100.0, 1500.0, n).astype(np.float32)
    periods = (50 + 3 * np.arange(n)).astype(np.float32)
    g = clocked_voice(wire)
    assert "EventPassthrough" not in oscen_amd.Graph(dsl=g.to_dsl(), per_voice=("frequency", "period")).kernel_source()
    close(render(g, freqs, periods, frames), model(freqs, periods, frames, gate_scale=0.5))


def test_event_fan_in_is_last_write_wins():
    """two sources into env.gate: only the last connected one delivers (clear + copy) -- here the graph's own gate input
    is connected first and the clock second, so scheduled gate events never reach the envelope"""
    def wire(g):
        g.input_event("gate")
        g.connect("gate", "env.gate")
        g.connect("clk.trig", "env.gate")

    n, frames = 64, 768
    freqs = np.linspace(100.0, 1500.0, n).astype(np.float32)
    periods = (41 + 7 * np.arange(n)).astype(np.float32)
    g = clocked_voice(wire)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_values("period", periods)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    for v in range(n):
        eng.schedule_voice_event("gate", v, 10 + v, 1.0)
    got = []
    for b in (256, 256, 256):
        eng.process_block(b)
        got.append(eng.read_voice_taps(b))
    close(np.concatenate(got, axis=1), model(freqs, periods, frames))


def test_passthrough_between_a_graph_event_input_and_a_node_is_free():
    """gate -> pass.input; pass.output -> env.gate is the kernel of gate -> env.gate (same hash, pipelines still on)"""
    def build(with_pass):
        g = oscen_amd.Graph("pt")
        g.input_value("frequency", 220.0, per_voice=True)
        g.input_event("gate")
        g.output_stream("out")
        g.node("env", "AdsrEnvelope::new", 0.002, 0.01, 0.6, 0.01)
        g.node("osc", "PolyBlepOscillator::saw", 220.0, 1.0)
        g.connect("frequency", "osc.frequency")
        if with_pass:
            g.node("pass", "EventPassthrough::new")
            g.connect("gate", "pass.input")
            g.connect("pass.output", "env.gate")
        else:
            g.connect("gate", "env.gate")
        g.connect("osc.output * env.output", "out")
        return g

    a, b = oscen_amd.Engine(build(True), 64, sample_rate=SR), oscen_amd.Engine(build(False), 64, sample_rate=SR)
    assert a.kernel_hash == b.kernel_hash
    for e in (a, b):
        for v in range(64):
            e.schedule_voice_event("gate", v, 3 * v, 0.9)
    assert np.array_equal(a.process_block(256), b.process_block(256))
