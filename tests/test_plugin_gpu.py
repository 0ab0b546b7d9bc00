"""GPU parity of the round-2 front-end surface: custom nodes registered through the plug-in API
(#[derive(Node)], oscen-macros/src/lib.rs:7-327), node arrays, nested graphs, stream inputs and
BlockRender::render(inputs, tail) (oscen-lib/src/graph/offline.rs:46-90)."""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import plugin_nodes
from tests.graph_interp import VoiceInterp
from tests.test_plugin_cpu import ARRAY_DSL, FLAT_DSL, INNER_DSL, OUTER_DSL

pytestmark = pytest.mark.gpu
SR = 48000.0


def drive(eng, n, total, block, ramps=True):
    """the same stimulus for every engine of a comparison: notes, a retrigger, ramped and plain parameter changes"""
    plans = oscen_amd.note_plans(n, span=total)
    oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    bus, taps = [], []
    for b in range(total // block):
        if ramps and b == 1:
            eng.set_value("op3_feedback", 0.12)
            eng.set_value("route", 0.4)
            eng.set_value_with_ramp("filter_cutoff", 5200.0, 700)
        if ramps and b == 2:
            eng.set_value("filter_env_amount", 1500.0)
            eng.set_value("op2_level", 0.7)
        bus.append(eng.process_block(block).copy())
        taps.append(eng.read_voice_taps(block))
    return np.concatenate(bus, axis=0), np.concatenate(taps, axis=1)


@pytest.mark.parametrize("n", [200, 40000])
def test_example_crate_nodes_through_the_plugin_api_match_the_builtin_kernel_bit_for_bit(n):
    """FmOperator / Crossfade / Mixer / AddValue are USER nodes in the reference (examples/fm-synth/src/nodes/*.rs).
    Registered through og_register_node and compiled by hiprtc they must give the bits of the built-in fm_voice
    kernel (same IEEE operations; the built-in only hoists the block-constant phase increment)."""
    total, block = 1024, 256
    ref_bus, ref_taps = drive(oscen_amd.Engine("fm_voice", n, sample_rate=SR), n, total, block)
    eng = oscen_amd.Engine(plugin_nodes.user_fm_voice(), n, sample_rate=SR)
    assert eng.lib.og_kernel_is_jit(eng.h) == 1
    bus, taps = drive(eng, n, total, block)
    assert np.array_equal(taps, ref_taps)
    assert np.array_equal(bus, ref_bus)
    assert np.abs(ref_taps).max() > 1e-2


def test_user_node_with_event_handler_and_integer_state():
    """a node with an event input, a u32 field and a constructor argument: a pluck -- on gate the counter restarts,
    the output is velocity * decay^counter * a square wave at the voice frequency"""
    oscen_amd.register_node(
        "Pluck::new",
        inputs=[("frequency", "value", 220.0, -1), ("gate", "event", 0.0, -1), ("decay", "value", 0.999, 0)],
        outputs=["output"], n_ctor_args=1,
        state=[("count", "u32", 0, -1), ("amp", "f32", 0.0, -1), ("phase", "f32", 0.0, -1)],
        process="""
    amp = amp * decay;
    count += 1u;
    output = (phase < 0.5f ? amp : -amp);
    const float p = phase + frequency / sample_rate;
    phase = p - truncf(p);
""",
        handlers={"gate": "    if (value > 0.0f) { amp = value; count = 0u; phase = 0.0f; }\n"})
    g = oscen_amd.Graph("plucks")
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_event("gate")
    g.output_stream("out")
    g.node("p", "Pluck::new", 0.9995)
    g.connect("frequency", "p.frequency")
    g.connect("gate", "p.gate")
    g.connect("p.output", "out")
    n, frames = 70, 600
    freqs = np.linspace(110.0, 1760.0, n).astype(np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    on = [(3 * v) % 400 for v in range(n)]
    for v in range(n):
        eng.schedule_voice_event("gate", v, on[v], 0.5 + 0.005 * v)
    got = []
    for _ in range(2):
        eng.process_block(frames // 2)
        got.append(eng.read_voice_taps(frames // 2))
    got = np.concatenate(got, axis=1)
    ref = np.zeros((n, frames), dtype=np.float32)
    f32 = np.float32
    for v in range(n):
        amp, phase = f32(0), f32(0)
        inc = f32(freqs[v] / f32(SR))
        for f in range(frames):
            if f == on[v]:
                amp, phase = f32(0.5 + 0.005 * v), f32(0)
            amp = f32(amp * f32(0.9995))
            ref[v, f] = amp if phase < f32(0.5) else -amp
            p = f32(phase + inc)
            phase = f32(p - np.trunc(p))
    assert np.array_equal(got, ref)


def test_sine_helpers_domain_of_the_hardware_sine_and_the_wide_form():
    """og_sin_turns is v_sin_f32 on the device.  The ISA documents it for |t| <= 256 turns (a deep modulation
    would silently give zeros); measured, gfx950 reduces any argument itself -- this test holds that.  og_sin_turns_wide
    takes the fractional part first and is a sine at any argument size whatever the part does."""
    oscen_amd.register_node(
        "SineProbe::new", inputs=[("frequency", "value", 1.0, -1)], outputs=["narrow", "wide"],
        state=[("k", "f32", 0.0, -1)],
        process="""
    const float t = frequency * (k + 1.0f) * 0.001f + 0.125f;
    narrow = og_sin_turns(t);
    wide = og_sin_turns_wide(t);
    k += 1.0f;
""")
    outs = {}
    n, frames = 64, 256
    scale = np.geomspace(1.0, 4.0e6, n).astype(np.float32)  # arguments up to ~1e6 turns
    for port in ("narrow", "wide"):
        g = oscen_amd.Graph("sine_probe_" + port)
        g.input_value("frequency", 1.0, per_voice=True)
        g.output_stream("out")
        g.node("p", "SineProbe::new")
        g.connect("frequency", "p.frequency")
        g.connect("p." + port, "out")
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        eng.set_voice_values("frequency", scale)
        eng.set_voice_taps(np.arange(n, dtype=np.uint32))
        eng.process_block(frames)
        outs[port] = eng.read_voice_taps(frames)
    k = np.arange(1, frames + 1, dtype=np.float32)[None, :]
    t = ((scale[:, None] * k).astype(np.float32) * np.float32(0.001)).astype(np.float32)
    t = (t + np.float32(0.125)).astype(np.float32)
    t64 = t.astype(np.float64)
    ref = np.sin(2.0 * np.pi * (t64 - np.floor(t64)))
    assert np.max(np.abs(t)) > 1.0e5
    assert np.max(np.abs(outs["wide"] - ref)) < 2e-6  # the hardware sine on an exact fractional part
    assert np.max(np.abs(outs["narrow"] - ref)) < 2e-6  # ... and so is the bare instruction on this part, far beyond 256 turns


def test_node_arrays_match_the_hand_expanded_graph_and_the_interpreter():
    n, frames, blocks = 6, 192, 4
    freqs = np.array([55.0, 110.0, 164.81, 220.0, 440.0, 659.25], dtype=np.float32)
    flat = ARRAY_DSL
    for i in range(3):
        flat = flat.replace("oscs[%d]" % i, "oscs__%d" % i)
    flat = flat.replace("oscs = [PolyBlepOscillator::saw(220.0, 0.3); 3];",
                        "".join("oscs__%d = PolyBlepOscillator::saw(220.0, 0.3);\n" % i for i in range(3)))
    flat = flat.replace("amps = [Gain::new(1.0); 3];", "".join("amps__%d = Gain::new(1.0);\n" % i for i in range(3)))
    flat = flat.replace("frequency -> oscs.frequency;", "".join("frequency -> oscs__%d.frequency;\n" % i for i in range(3)))
    flat = flat.replace("oscs.output -> amps.input;", "".join("oscs__%d.output -> amps__%d.input;\n" % (i, i) for i in range(3)))
    flat = flat.replace("env.output -> amps.gain;", "".join("env.output -> amps__%d.gain;\n" % i for i in range(3)))
    flat = flat.replace("amps.output -> filter.input;", "".join("amps__%d.output -> filter.input;\n" % i for i in range(3)))
    outs = []
    for dsl in (ARRAY_DSL, flat):
        eng = oscen_amd.Engine(oscen_amd.Graph(dsl=dsl, per_voice=("frequency",)), n, sample_rate=SR)
        eng.set_voice_values("frequency", freqs)
        eng.set_voice_taps(np.arange(n, dtype=np.uint32))
        for v in range(n):
            eng.schedule_voice_event("gate", v, 5 + 7 * v, 0.8)
            eng.schedule_voice_event("gate", v, 400 + 3 * v, 0.0)
        t = []
        for b in range(blocks):
            if b == 2:
                eng.set_value("detune", 0.03)
            eng.process_block(frames)
            t.append(eng.read_voice_taps(frames))
        outs.append(np.concatenate(t, axis=1))
    assert np.array_equal(outs[0], outs[1])
    # and against the per-sample interpreter over the oracle's nodes
    desc = {
        "inputs": [("frequency", "value", 220.0, 0), ("detune", "value", 0.01, 0), ("gate", "event", 0.0, 0)],
        "nodes": [("env", "AdsrEnvelope::new", [0.01, 0.1, 0.7, 0.2])] +
                 [("oscs__%d" % i, "PolyBlepOscillator::saw", [220.0, 0.3]) for i in range(3)] +
                 [("amps__%d" % i, "Gain::new", [1.0]) for i in range(3)] + [("filter", "TptFilter::new", [2400.0, 0.8])],
        "edges": [("gate", "env.gate")] + [("frequency", "oscs__%d.frequency" % i) for i in range(3)] +
                 [("detune * 2.0", "oscs__2.frequency_mod"), ("detune", "oscs__1.frequency_mod")] +
                 [("oscs__%d.output" % i, "amps__%d.input" % i) for i in range(3)] +
                 [("env.output", "amps__%d.gain" % i) for i in range(3)] +
                 [("amps__%d.output" % i, "filter.input") for i in range(3)] + [("filter.output", "out")],
        "order": ["env", "oscs__0", "oscs__1", "oscs__2", "amps__0", "amps__1", "amps__2", "filter"],
    }
    ref = np.zeros((n, frames * blocks), dtype=np.float32)
    for v in range(n):
        vi = VoiceInterp(desc, SR, {"frequency": float(freqs[v])})
        for f in range(frames * blocks):
            if f == 2 * frames:
                vi.set_value("detune", 0.03)
            gates = [("gate", 0.8)] if f == 5 + 7 * v else ([("gate", 0.0)] if f == 400 + 3 * v else [])
            ref[v, f] = vi.frame(gates)
    err = float(np.max(np.abs(outs[0] - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5 and np.abs(ref).max() > 1e-2, err


def test_nested_graph_equals_the_flat_graph():
    oscen_amd.register_graph_type("OscEnv", oscen_amd.Graph(dsl=INNER_DSL))
    try:
        outs = []
        for dsl in (OUTER_DSL, FLAT_DSL):
            n = 9
            eng = oscen_amd.Engine(oscen_amd.Graph(dsl=dsl, per_voice=("frequency",)), n, sample_rate=SR)
            eng.set_voice_values("frequency", np.linspace(55.0, 880.0, n).astype(np.float32))
            eng.set_voice_taps(np.arange(n, dtype=np.uint32))
            for v in range(n):
                eng.schedule_voice_event("gate", v, 11 * v, 0.9)
                eng.schedule_voice_event("gate", v, 300 + v, 0.0)
            t = []
            for _ in range(3):
                eng.process_block(200)
                t.append(eng.read_voice_taps(200))
            outs.append(np.concatenate(t, axis=1))
        assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0]).max() > 1e-2
    finally:
        oscen_amd.unregister_graph_type("OscEnv")


def fx_graph():
    g = oscen_amd.Graph("fx")
    g.input_stream("audio_in")
    g.input_value("cutoff", 1200.0, per_voice=True)
    g.input_value("drive", 1.0)
    g.output_stream("out")
    g.node("f", "TptFilter::new", 1200.0, 0.9)
    g.node("g", "Gain::new", 1.0)
    g.connect("audio_in * drive", "f.input")
    g.connect("cutoff", "f.cutoff")
    g.connect("f.output", "g.input")
    g.connect("audio_in * 0.25 + 0.5", "g.gain")
    g.connect("g.output", "out")
    return g


def test_stream_input_block_and_render_with_inputs():
    """`<stream_in>_block` feeds every voice of the bank; render(inputs, tail) = chunks of 512 with silence padding."""
    n = 5
    cut = np.array([300.0, 800.0, 1500.0, 4000.0, 9000.0], dtype=np.float32)
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(1300) * 0.4).astype(np.float32)
    tail = 200
    total = len(x) + tail
    # (a) render(inputs, tail)
    eng = oscen_amd.Engine(fx_graph(), n, sample_rate=SR)
    assert eng.lib.og_num_stream_inputs(eng.h) == 1
    eng.set_voice_values("cutoff", cut)
    eng.set_value("drive", 0.9)
    bus = eng.render_inputs([x], tail=tail)
    assert bus.shape == (total, 1)
    # (b) the same through set_stream_block + process_block with ragged blocks, voices tapped
    eng2 = oscen_amd.Engine(fx_graph(), n, sample_rate=SR)
    eng2.set_voice_values("cutoff", cut)
    eng2.set_value("drive", 0.9)
    eng2.set_voice_taps(np.arange(n, dtype=np.uint32))
    xp = np.concatenate([x, np.zeros(tail, dtype=np.float32)])
    taps, bus2, pos = [], [], 0
    for blk in (512, 512, 100, 376):
        eng2.set_stream_block("audio_in", xp[pos:pos + blk])
        bus2.append(eng2.process_block(blk).copy())
        taps.append(eng2.read_voice_taps(blk))
        pos += blk
    assert pos == total
    taps = np.concatenate(taps, axis=1)
    bus2 = np.concatenate(bus2, axis=0)
    assert np.max(np.abs(bus2 - bus)) <= 1e-6 * max(1.0, float(np.abs(bus).max()))  # 512-frame vs ragged chunks: same samples
    # (c) against the interpreter: a stream input is a value that changes every frame
    desc = {"inputs": [("audio_in", "value", 0.0, 0), ("cutoff", "value", 1200.0, 0), ("drive", "value", 1.0, 0)],
            "nodes": [("f", "TptFilter::new", [1200.0, 0.9]), ("g", "Gain::new", [1.0])],
            "edges": [("audio_in * drive", "f.input"), ("cutoff", "f.cutoff"), ("f.output", "g.input"),
                      ("audio_in * 0.25 + 0.5", "g.gain"), ("g.output", "out")],
            "order": ["f", "g"]}
    ref = np.zeros((n, total), dtype=np.float32)
    for v in range(n):
        vi = VoiceInterp(desc, SR, {"cutoff": float(cut[v])})
        vi.set_value("drive", 0.9)
        for f in range(total):
            vi.set_value("audio_in", float(xp[f]))
            ref[v, f] = vi.frame([])
    err = float(np.max(np.abs(taps - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5 and np.abs(ref).max() > 1e-2, err
    want = ref.astype(np.float64).sum(axis=0)
    assert np.max(np.abs(bus[:, 0] - want)) <= 1e-5 * max(1.0, float(np.abs(want).max()))
    # render() insists on NUM_STREAM_INPUTS buffers
    with pytest.raises(oscen_amd.OscenError):
        eng.render_inputs([], tail=10)
