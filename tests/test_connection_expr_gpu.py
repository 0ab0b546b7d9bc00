"""Calls on a connection, on the GPU (ast.rs:113-129 `ConnectionExpr::{Call, MethodCall, ArrayIndex}`):

* named pure functions -- frame -> frame, scalar -> scalar, several arguments -> frame, path-qualified, broadcast into
  node arrays -- with the known answers of the reference's own tests (oscen-lib/tests/connection_expr_functions.rs:136-330,
  connection_expr_function_paths.rs:192-262): mid/side (0.6, 0.1) -> (0.5, 0.7), half(0.8) = 0.4, merge2(0.25, -0.5);
* the frame constructor in its three spellings and channel extraction (connection_expr_frames.rs:104-131);
* f32 methods (`x.tanh()`, `x.clamp(lo, hi)`, ...: the reference passes them through to Rust's f32) against numpy's f32
  functions on a per-voice sweep, tolerance 1e-5 * max(1, |ref|) (BASELINE.json north_star);
* a method on a block-uniform value (evaluated on the host, so the consumer's coefficients stay hoisted).
"""
import numpy as np
import pytest

import oscen_amd
from tests import observed

pytestmark = pytest.mark.gpu
SR = 48000.0
f32 = np.float32

NODES = {
    "CxConstF32::new": dict(inputs=[], outputs=["output"], n_ctor_args=1, state=[("val", "f32", 0.0, 0)], process="    output = val;\n"),
    "CxStereoConst::new": dict(inputs=[], outputs=[("output", 2)], n_ctor_args=2, state=[("l", "f32", 0.0, 0), ("r", "f32", 0.0, 1)],
                               process="    output.v[0] = l;\n    output.v[1] = r;\n"),
    "CxStereoSink::new": dict(inputs=[("input", "stream", 0.0, -1, 2)], outputs=[("last", 2)], process="    last = input;\n"),
    "CxMonoSink::new": dict(inputs=[("input", "stream", 0.0, -1)], outputs=["last"], process="    last = input;\n"),
}
FUNCS = {
    "decode_ms": dict(args=[("v", 2)], result_channels=2,
                      source="og::Frame<2> o; o.v[0] = v.v[0] - v.v[1]; o.v[1] = v.v[0] + v.v[1]; return o;"),
    "half": dict(args=["x"], source="return x * 0.5f;"),
    "merge2": dict(args=["l", "r"], result_channels=2, source="og::Frame<2> o; o.v[0] = l; o.v[1] = r; return o;"),
}


@pytest.fixture()
def registered():
    for k, v in NODES.items():
        oscen_amd.register_node(k, **v)
    for k, v in FUNCS.items():
        oscen_amd.register_function(k, **v)
    yield
    for k in NODES:
        oscen_amd.unregister_node(k)
    for k in FUNCS:
        oscen_amd.unregister_function(k)


def bus_of(text, n=3, frames=16):
    eng = oscen_amd.Engine(oscen_amd.Graph(dsl=text), n, sample_rate=SR)
    return eng.process_block(frames) / f32(n)  # every voice holds the same constants: the bus is n times one voice


def test_named_functions_with_the_reference_known_answers(registered):
    cases = [
        ("output out: stream: Frame<2>;", "s = CxStereoConst::new(0.6, 0.1);", "decode_ms(s.output) -> out;", (0.5, 0.7)),
        ("output out: stream: Frame<2>;", "s = CxStereoConst::new(0.6, 0.1);", "dsp::decode_ms(s.output) -> out;", (0.5, 0.7)),
        ("output out: stream;", "a = CxConstF32::new(0.8);", "half(a.output) -> out;", (0.4,)),
        ("output out: stream: Frame<2>;", "a = CxConstF32::new(0.25); b = CxConstF32::new(-0.5);", "merge2(a.output, b.output) -> out;", (0.25, -0.5)),
        # the frame constructor: bare, turbofish, path-qualified (any call path ending in `Frame`)
        ("output out: stream: Frame<2>;", "a = CxConstF32::new(0.25); b = CxConstF32::new(-0.5);", "Frame(a.output, b.output) -> out;", (0.25, -0.5)),
        ("output out: stream: Frame<2>;", "a = CxConstF32::new(0.25); b = CxConstF32::new(-0.5);", "Frame::<2>(a.output, b.output) -> out;", (0.25, -0.5)),
        ("output out: stream: Frame<2>;", "a = CxConstF32::new(0.25); b = CxConstF32::new(-0.5);",
         "oscen::frame::Frame::<2>(a.output, b.output) -> out;", (0.25, -0.5)),
        # functions compose with the arithmetic of a compound source and with each other
        ("output out: stream;", "a = CxConstF32::new(0.8); b = CxConstF32::new(-0.5);", "half(a.output + b.output) * 4.0 - half(half(b.output)) -> out;",
         (float(f32(f32(f32(0.8) + f32(-0.5)) * f32(0.5)) * f32(4.0) - f32(f32(-0.5) * f32(0.5)) * f32(0.5)),)),
        ("output out: stream;", "s = CxStereoConst::new(0.6, 0.1);", "decode_ms(s.output)[1] - decode_ms(s.output)[0] -> out;", (0.2,)),
    ]
    for out_decl, nodes, conn, want in cases:
        bus = bus_of(f"name: CxG; {out_decl} nodes {{ {nodes} }} connections {{ {conn} }}")
        assert bus.shape[1] == len(want), conn
        assert np.allclose(bus, np.array(want, dtype=f32)[None, :], rtol=0, atol=1e-6), (conn, bus[0])


def test_channel_extraction_and_broadcast_into_node_arrays(registered):
    # connection_expr_frames.rs:118-131: each channel of a Frame<2> source routed to its own mono sink
    bus = bus_of("""name: CxExtract; output l: stream; output r: stream;
        nodes { s = CxStereoConst::new(0.3, -0.7); left = CxMonoSink::new(); right = CxMonoSink::new(); }
        connections { s.output[0] -> left.input; s.output[1] -> right.input; left.last -> l; right.last * 2.0 -> r; }""")
    assert np.allclose(bus, np.array([0.3, -1.4], dtype=f32)[None, :], atol=1e-6)
    # connection_expr_functions.rs:232-277: a frame function, a scalar function and a frame constructor fan into arrays
    # (every element receives the value: the sum over the three sinks is three times it)
    for nodes, conn, sink, want in [
        ("s = CxStereoConst::new(0.6, 0.1); sinks = [CxStereoSink::new(); 3];", "decode_ms(s.output) -> sinks.input;", "sinks", (1.5, 2.1)),
        ("s = CxStereoConst::new(0.6, 0.1); sinks = [CxStereoSink::new(); 3];", "dsp::decode_ms(s.output) -> sinks.input;", "sinks", (1.5, 2.1)),
        ("a = CxConstF32::new(0.8); sinks = [CxMonoSink::new(); 3];", "half(a.output) -> sinks.input;", "sinks", (1.2,)),
        ("a = CxConstF32::new(0.25); b = CxConstF32::new(-0.5); sinks = [CxStereoSink::new(); 3];",
         "Frame::<2>(a.output, b.output) -> sinks.input;", "sinks", (0.75, -1.5)),
        ("a = CxConstF32::new(0.25); sinks = [CxMonoSink::new(); 3];", "a.output * 2.0 -> sinks.input;", "sinks", (1.5,)),
    ]:
        ty = "stream: Frame<2>" if len(want) == 2 else "stream"
        bus = bus_of(f"name: CxB; output out: {ty}; nodes {{ {nodes} }} connections {{ {conn} {sink}.last -> out; }}")
        assert np.allclose(bus, np.array(want, dtype=f32)[None, :], atol=1e-6), (conn, bus[0])
    # array_frame_composition.rs:47-59,96-108: an array of Frame<2> sources sums per channel into a Frame<2> output
    g = oscen_amd.Graph(dsl="""name: CxSum; output out: stream: Frame<2>;
        nodes { v0 = CxStereoConst::new(0.1, -0.2); v1 = CxStereoConst::new(0.4, 0.5); v2 = CxStereoConst::new(-0.05, 0.7); }
        connections { v0.output -> out; v1.output -> out; v2.output -> out; }""")
    bus = oscen_amd.Engine(g, 1, sample_rate=SR).process_block(8)
    assert np.allclose(bus, np.array([0.45, 1.0], dtype=f32)[None, :], atol=1e-6)


def _round_half_away(x):
    return np.trunc(x + np.copysign(f32(0.5), x)).astype(f32)


METHODS = [  # (expression over the per-voice value x and the uniform value y, numpy f32 reference)
    ("x.abs()", lambda x, y: np.abs(x)),
    ("x.abs().sqrt()", lambda x, y: np.sqrt(np.abs(x))),
    ("x.cbrt()", lambda x, y: np.cbrt(x)),
    ("(x + 4.0).recip()", lambda x, y: f32(1.0) / (x + f32(4.0))),
    ("x.tanh()", lambda x, y: np.tanh(x)),
    ("x.sinh()", lambda x, y: np.sinh(x)),
    ("x.cosh()", lambda x, y: np.cosh(x)),
    ("x.sin()", lambda x, y: np.sin(x)),
    ("x.cos()", lambda x, y: np.cos(x)),
    ("(x * 0.4).tan()", lambda x, y: np.tan(x * f32(0.4))),
    ("(x * 0.3).asin()", lambda x, y: np.arcsin(x * f32(0.3))),
    ("(x * 0.3).acos()", lambda x, y: np.arccos(x * f32(0.3))),
    ("x.atan()", lambda x, y: np.arctan(x)),
    ("x.atan2(y)", lambda x, y: np.arctan2(x, y)),
    ("x.hypot(y)", lambda x, y: np.hypot(x, y)),
    ("x.exp()", lambda x, y: np.exp(x)),
    ("x.exp2()", lambda x, y: np.exp2(x)),
    ("x.exp_m1()", lambda x, y: np.expm1(x)),
    ("(x.abs() + 0.1).ln()", lambda x, y: np.log(np.abs(x) + f32(0.1))),
    ("x.abs().ln_1p()", lambda x, y: np.log1p(np.abs(x))),
    ("(x.abs() + 0.1).log2()", lambda x, y: np.log2(np.abs(x) + f32(0.1))),
    ("(x.abs() + 0.1).log10()", lambda x, y: np.log10(np.abs(x) + f32(0.1))),
    ("x.abs().powf(1.7)", lambda x, y: np.power(np.abs(x), f32(1.7))),
    ("x.powi(3)", lambda x, y: x * x * x),
    ("(x * 2.3).floor()", lambda x, y: np.floor(x * f32(2.3))),
    ("(x * 2.3).ceil()", lambda x, y: np.ceil(x * f32(2.3))),
    ("(x * 2.5).round()", lambda x, y: _round_half_away(x * f32(2.5))),
    ("(x * 2.3).trunc()", lambda x, y: np.trunc(x * f32(2.3))),
    ("(x * 2.3).fract()", lambda x, y: x * f32(2.3) - np.trunc(x * f32(2.3))),
    ("x.signum()", lambda x, y: np.where(np.signbit(x), f32(-1.0), f32(1.0)).astype(f32)),
    ("x.min(y)", lambda x, y: np.minimum(x, y)),
    ("x.max(y)", lambda x, y: np.maximum(x, y)),
    ("x.clamp(-0.75, 1.25)", lambda x, y: np.clip(x, f32(-0.75), f32(1.25))),
    ("x.mul_add(y, 0.25)", lambda x, y: (x.astype(np.float64) * np.float64(y) + 0.25).astype(f32)),
    ("x.to_radians()", lambda x, y: np.radians(x)),
    ("x.to_degrees()", lambda x, y: np.degrees(x)),
]


def test_f32_methods_against_numpy_on_a_per_voice_sweep():
    n, frames = 257, 8
    x = np.linspace(-3.0, 3.0, n).astype(f32)
    x[n // 2] = f32(0.0)
    y = f32(0.7)
    worst = {}
    for k in range(0, len(METHODS), 4):  # four stream outputs (= bus channels) per graph
        group = METHODS[k:k + 4]
        outs = "".join(f"output o{i}: stream; " for i in range(len(group)))
        conns = "".join(f"{expr} -> o{i}; " for i, (expr, _) in enumerate(group))
        g = oscen_amd.Graph(dsl=f"name: CxM{k}; input x: value = 0.0; input y: value = 0.7; {outs} nodes {{ }} connections {{ {conns} }}",
                            per_voice=["x"])
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        eng.set_voice_values("x", x)
        eng.set_voice_taps(list(range(n)))
        eng.process_block(frames)
        taps = eng.read_voice_taps(frames)  # (voice, frame, channel)
        for i, (expr, ref) in enumerate(group):
            want = np.asarray(ref(x, y), dtype=f32)
            got = taps[:, -1, i]
            err = float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))
            worst[expr] = err
            observed.note(err)
            assert err <= 1e-5, (expr, err, got[:4], want[:4])
            assert np.array_equal(taps[:, 0, i], got)  # constant over the block
    assert len(worst) == len(METHODS)


def test_method_on_a_block_uniform_value_is_evaluated_on_the_host():
    """`cutoff.clamp(200.0, 2000.0) -> filter.cutoff` with `cutoff` a broadcast value input: the clamped value stays
    block-uniform (the TPT coefficients are then computed once per block, not per sample) and follows set_value"""
    def build(conn):
        return oscen_amd.Graph(dsl=f"""name: CxU; input cutoff: value = 9000.0; input frequency: value = 220.0; output out: stream;
            nodes {{ osc = PolyBlepOscillator::saw(220.0, 0.5); filter = TptFilter::new(1000.0, 0.7); }}
            connections {{ frequency -> osc.frequency; osc.output -> filter.input; {conn} filter.output -> out; }}""", per_voice=["frequency"])
    a, b = build("cutoff.clamp(200.0, 2000.0) -> filter.cutoff;"), build("cutoff -> filter.cutoff;")
    assert "tpt_params_nomod(" in a.kernel_source()  # the host-computed coefficient form: the clamped cutoff is block-uniform
    n, frames = 5, 128
    freqs = np.array([55.0, 110.0, 220.0, 440.0, 880.0], dtype=f32)
    outs = []
    for g, cut in ((a, 9000.0), (b, 2000.0), (a, 50.0), (b, 200.0), (a, 700.0), (b, 700.0)):
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        eng.set_voice_values("frequency", freqs)
        eng.set_value_immediate("cutoff", cut)
        eng.set_voice_taps(list(range(n)))
        eng.process_block(frames)
        outs.append(eng.read_voice_taps(frames))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[2], outs[3]) and np.array_equal(outs[4], outs[5])
    assert np.abs(outs[0]).max() > 0.05 and not np.array_equal(outs[0], outs[2])


def test_output_less_fixture_graphs_read_through_node_fields(registered):
    """the reference's sink fixtures declare no graph output and are read through the nodes' public fields
    (`graph.sinks[i].last`, connection_expr_functions.rs:232-330; ir/passes/dead_nodes.rs:17-19 keeps every node of such a
    graph): og_read_state_field plays that part"""
    oscen_amd.register_node("CxLatch::new", inputs=[("input", "stream", 0.0, -1)], outputs=[], state=[("last", "f32", 0.0, -1)],
                            process="    last = input;\n")
    oscen_amd.register_node("CxLatch2::new", inputs=[("input", "stream", 0.0, -1, 2)], outputs=[],
                            state=[("last_l", "f32", 0.0, -1), ("last_r", "f32", 0.0, -1)], process="    last_l = input.v[0];\n    last_r = input.v[1];\n")
    try:
        n = 70
        x = np.linspace(-1.0, 1.0, n).astype(f32)
        g = oscen_amd.Graph(dsl="""name: CxNoOut; input x: value = 0.0;
            nodes { s = CxStereoConst::new(0.6, 0.1); sinks = [CxLatch2::new(); 3]; mono = [CxLatch::new(); 2]; one = CxLatch::new(); }
            connections { decode_ms(s.output) -> sinks.input; half(x) -> mono.input; x.abs() * 3.0 -> one.input; }""", per_voice=["x"])
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        eng.set_voice_values("x", x)
        bus = eng.process_block(32)
        assert not bus.any()  # no graph output: a silent bus
        for i in range(3):
            assert np.allclose(eng.read_state_field("sinks[%d].last_l" % i), 0.5, atol=1e-6)
            assert np.allclose(eng.read_state_field("sinks[%d].last_r" % i), 0.7, atol=1e-6)
        for i in range(2):
            assert np.array_equal(eng.read_state_field("mono[%d].last" % i), x * f32(0.5))
        assert np.array_equal(eng.read_state_field("one.last", first=10, n=20), (np.abs(x) * f32(3.0))[10:30])
        assert eng.lib.og_state_field_index(eng.h, b"one.last") >= 0 and eng.lib.og_state_field_index(eng.h, b"one.nothing") == -1
        with pytest.raises(oscen_amd.OscenError):
            eng.read_state_field("nobody.last")
    finally:
        oscen_amd.unregister_node("CxLatch::new")
        oscen_amd.unregister_node("CxLatch2::new")
