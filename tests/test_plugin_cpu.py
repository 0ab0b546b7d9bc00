"""CPU checks of the graph front end's round-2 surface: custom node types (og_register_node = #[derive(Node)]),
node arrays, nested graphs and stream inputs lower to a kernel that compiles for gfx950 (hiprtc, no device)."""
import pytest

import oscen_amd
from tests import plugin_nodes


def test_example_crate_nodes_as_plugins_compile_into_the_voice_kernel():
    g = plugin_nodes.user_fm_voice()
    src = g.kernel_source()
    for fn in ("og_user_UFmOperator__new_process", "og_user_UCrossfade__new_process", "og_user_UMixer__new_process",
               "og_user_UAddValue__new_process"):
        assert src.count(fn + "(") >= 2  # definition + at least one call
    # one definition + three operators in each of the three kernel bodies (ordinary, two-wave and four-wave pipelines)
    assert src.count("og_user_UFmOperator__new_process(") == 1 + 3 * 3
    assert g.jit_check("gfx950") > 10000
    # a built-in name cannot be shadowed, malformed descriptions are rejected
    with pytest.raises(oscen_amd.OscenError):
        oscen_amd.register_node("Gain::new", [("input", "stream", 0.0, -1)], ["output"], "output = input;")
    with pytest.raises(oscen_amd.OscenError):
        oscen_amd.register_node("Bad::new", [("in put", "stream", 0.0, -1)], ["output"], "output = 0.0f;")
    with pytest.raises(oscen_amd.OscenError):
        oscen_amd.register_node("Bad::new", [("x", "stream", 0.0, 3)], ["output"], "output = x;")
    # unknown types point at the plug-in API
    u = oscen_amd.Graph("u")
    u.output_stream("out")
    u.node("n", "Nope::new")
    u.connect("n.output", "out")
    with pytest.raises(oscen_amd.OscenError, match="og_register_node"):
        u.kernel_source()


ARRAY_DSL = """
name: Unison;
input frequency: value = 220.0;
input detune: value = 0.01;
input gate: event;
output out: stream;
nodes {
    env = AdsrEnvelope::new(0.01, 0.1, 0.7, 0.2);
    oscs = [PolyBlepOscillator::saw(220.0, 0.3); 3];
    amps = [Gain::new(1.0); 3];
    filter = TptFilter::new(2400.0, 0.8);
}
connections {
    gate -> env.gate;
    frequency -> oscs.frequency;            // broadcast: scalar -> array
    detune * 2.0 -> oscs[2].frequency_mod;  // one element
    detune -> oscs[1].frequency_mod;
    oscs.output -> amps.input;              // parallel: array -> array
    env.output -> amps.gain;                // broadcast
    amps.output -> filter.input;            // fan-in: sum in index order
    filter.output -> out;
}
"""


def test_node_arrays_expand_with_the_reference_fanout_rules():
    g = oscen_amd.Graph(dsl=ARRAY_DSL, per_voice=("frequency",))
    assert "oscs = [PolyBlepOscillator::saw(220.0, 0.300000012); 3];" in g.to_dsl()
    src = g.kernel_source()
    order = [ln for ln in src.splitlines() if ln.startswith("// Node order:")][0]
    assert order.split()[3:] == ["env", "oscs__0", "oscs__1", "oscs__2", "amps__0", "amps__1", "amps__2", "filter"]
    # the fan-in is ((amps0 + amps1) + amps2), in index order
    assert "((x4_n4_output + x5_n5_output) + x6_n6_output)" in src
    assert g.jit_check("gfx950") > 10000
    # round trip through the DSL printer
    g2 = oscen_amd.Graph(dsl=g.to_dsl(), per_voice=("frequency",))
    assert g2.kernel_source() == src
    # errors the reference also diagnoses
    bad = oscen_amd.Graph(dsl=ARRAY_DSL.replace("amps = [Gain::new(1.0); 3];", "amps = [Gain::new(1.0); 2];"),
                          per_voice=("frequency",))
    with pytest.raises(oscen_amd.OscenError, match="lengths differ"):
        bad.kernel_source()
    bad = oscen_amd.Graph(dsl=ARRAY_DSL.replace("amps.output -> filter.input;", "amps.output * 0.5 -> filter.input;"),
                          per_voice=("frequency",))
    with pytest.raises(oscen_amd.OscenError, match="compound expression"):
        bad.kernel_source()
    bad = oscen_amd.Graph(dsl=ARRAY_DSL.replace("oscs[2]", "oscs[3]"), per_voice=("frequency",))
    with pytest.raises(oscen_amd.OscenError, match="out of range"):
        bad.kernel_source()


INNER_DSL = """
name: OscEnv;
input frequency: value = 110.0;
input level: value = 0.5;
input gate: event;
output out: stream;
nodes {
    env = AdsrEnvelope::new(0.005, 0.05, 0.6, 0.1);
    osc = PolyBlepOscillator::square(110.0, 0.4);
}
connections {
    gate -> env.gate;
    frequency -> osc.frequency;
    osc.output * env.output * level -> out;
}
"""
OUTER_DSL = """
name: TwoLayers;
input frequency: value = 110.0;
input gate: event;
output out: stream;
nodes {
    low = OscEnv;
    high = OscEnv::new();
    filter = TptFilter::new(1800.0, 0.7);
}
connections {
    gate -> low.gate;
    gate -> high.gate;
    frequency -> low.frequency;
    frequency * 2.0 -> high.frequency;
    low.out + high.out * 0.5 -> filter.input;
    filter.output -> out;
}
"""
FLAT_DSL = """
name: TwoLayers;
input frequency: value = 110.0;
input gate: event;
output out: stream;
nodes {
    low_env = AdsrEnvelope::new(0.005, 0.05, 0.6, 0.1);
    low_osc = PolyBlepOscillator::square(110.0, 0.4);
    high_env = AdsrEnvelope::new(0.005, 0.05, 0.6, 0.1);
    high_osc = PolyBlepOscillator::square(110.0, 0.4);
    filter = TptFilter::new(1800.0, 0.7);
}
connections {
    gate -> low_env.gate;
    (frequency) -> low_osc.frequency;
    gate -> high_env.gate;
    (frequency * 2.0) -> high_osc.frequency;
    ((low_osc.output * low_env.output * (0.5))) + ((high_osc.output * high_env.output * (0.5))) * 0.5 -> filter.input;
    filter.output -> out;
}
"""


def test_nested_graphs_are_inlined():
    inner = oscen_amd.Graph(dsl=INNER_DSL)
    oscen_amd.register_graph_type("OscEnv", inner)
    try:
        nested = oscen_amd.Graph(dsl=OUTER_DSL, per_voice=("frequency",))
        flat = oscen_amd.Graph(dsl=FLAT_DSL, per_voice=("frequency",))
        assert nested.kernel_source() == flat.kernel_source()  # same nodes, same order, same expressions
        assert nested.jit_check("gfx950") > 10000
    finally:
        oscen_amd.unregister_graph_type("OscEnv")
    with pytest.raises(oscen_amd.OscenError, match="unknown node type"):
        oscen_amd.Graph(dsl=OUTER_DSL, per_voice=("frequency",)).kernel_source()


def test_stream_inputs_lower_to_the_per_frame_table():
    g = oscen_amd.Graph("fx")
    g.input_stream("audio_in")
    g.input_value("cutoff", 1200.0, per_voice=True)
    g.output_stream("out")
    g.node("f", "TptFilter::new", 1200.0, 0.9)
    g.node("clip", "HardClip::new", rate=4)
    g.connect("audio_in * 0.8", "f.input")
    g.connect("cutoff", "f.cutoff")
    g.connect("f.output + audio_in", "clip.input")  # outer -> oversampled: the stream is upsampled (sinc) like a node output
    g.connect("clip.output", "out")
    src = g.kernel_source()
    assert "ST(0)" in src and "og::sinc_up<4>" in src and "og::sinc_down<4>" in src
    assert g.jit_check("gfx950") > 10000


def test_event_outputs_lower_to_in_kernel_event_edges():
    """`#[output(event)]` fields: the producer's queue lives in registers, the consumer's handler runs in the tick right
    before its process(); EventPassthrough is removed; graph-level event outputs are refused."""
    oscen_amd.register_node(
        "Ticker::new", inputs=[("period", "value", 64.0, 0)], outputs=[], n_ctor_args=1, state=[("count", "u32", 0, -1)],
        event_outputs=["trig"], process="    count += 1u;\n    if ((float)count >= period) { count = 0u; trig.push(1.0f); }\n")
    g = oscen_amd.Graph("ev")
    g.input_event("gate")
    g.output_stream("out")
    g.node("t", "Ticker::new", 32.0)
    g.node("p", "EventPassthrough::new")
    g.node("env", "AdsrEnvelope::new", 0.01, 0.1, 0.5, 0.2)
    g.connect("gate", "env.gate")          # replaced by the later edge into the same input (clear + copy)
    g.connect("t.trig", "p.input")
    g.connect("p.output", "env.gate")
    g.connect("env.output", "out")
    src = g.kernel_source()
    assert "og::EvOut n0_trig;" in src and "og::adsr_gate(n1_e, evv, A" in src and "n0_trig.clear();" in src
    assert "EventPassthrough" not in src and "og_k2_" not in src      # no pipeline kernels for node-to-node events
    assert "ev.target == 0u" not in src                                 # the graph's gate input no longer reaches env
    assert g.jit_check() > 0
    bad = oscen_amd.Graph("ev2")
    bad.output_stream("out")
    bad.node("t", "Ticker::new", 32.0)
    bad.connect("t.trig", "out")
    with pytest.raises(oscen_amd.OscenError, match="is a stream output: an event source cannot feed it"):
        bad.kernel_source()
    # round 3: an EVENT output of the graph takes the node's events out of the voice (device log, og_read_output_events)
    ok = oscen_amd.Graph("ev3")
    ok.output_stream("out")
    ok.output_event("ticks")
    ok.node("t", "Ticker::new", 32.0)
    ok.node("osc", "Oscillator::sine", 220.0, 0.5)
    ok.connect("osc.output", "out")
    ok.connect("t.trig", "ticks")
    src = ok.kernel_source()
    assert "og::ev_out_log(A, c, 0u, f, n0_trig);" in src and "og::ev_report_lost(A, c, n0_trig);" in src
    assert src.index("og::ev_out_log(") < src.index("n0_trig.clear();") and "og_k2_" not in src
    assert ok.jit_check() > 0
    oscen_amd.unregister_node("Ticker::new")


def test_frame_ports_lower_to_channel_values():
    """Frame<N> edges: the channels travel as scalar values (so they cross pipeline cuts like any other value), user code
    sees og::Frame<N>; width mismatches are compile-time errors; a Frame<2> at the graph output makes the mix bus stereo."""
    oscen_amd.register_node("Wide::new", inputs=[("input", "stream", 0.0, -1)], outputs=[("output", 2)],
                            process="    output = og::Frame<2>::splat(input);\n")
    oscen_amd.register_node("Narrow::new", inputs=[("input", "stream", 0.0, -1, 2)], outputs=["output"],
                            process="    output = input.v[0] + input.v[1];\n")
    g = oscen_amd.Graph("fr")
    g.output_stream("out")
    g.node("osc", "Oscillator::sine", 330.0, 1.0)
    g.node("w", "Wide::new")
    g.node("f", "TptFilter::<Frame<2>>::new", 900.0, 0.7)
    g.node("n", "Narrow::new")
    g.connect("osc.output", "w.input")
    g.connect("w.output * osc.output + -w.output", "f.input")  # (ONE compound source: the reference refuses compound sources in a summed fan-in)
    g.connect("f.output", "n.input")
    g.connect("n.output", "out")
    assert "TptFilter::<Frame<2>>::new(900.0" in g.to_dsl()
    src = g.kernel_source()
    assert "og::Frame<2>& output" in src and "const og::Frame<2> input" in src
    assert "n2_z0_0" in src and "n2_z1_1" in src  # one integrator pair per channel
    assert g.jit_check() > 0

    def bad(wire, msg):
        h = oscen_amd.Graph("bad")
        h.output_stream("out")
        h.node("osc", "Oscillator::sine", 330.0, 1.0)
        h.node("w", "Wide::new")
        h.node("f", "TptFilter::<Frame<2>>::new", 900.0, 0.7)
        h.node("m", "TptFilter::new", 900.0, 0.7)
        h.node("n", "Narrow::new")
        h.connect("osc.output", "w.input")
        wire(h)
        with pytest.raises(oscen_amd.OscenError, match=msg):
            h.kernel_source()

    bad(lambda h: (h.connect("w.output", "m.input"), h.connect("m.output", "out")), "takes an f32 stream")
    bad(lambda h: (h.connect("osc.output", "f.input"), h.connect("f.output", "n.input"), h.connect("n.output", "out")),
        "its source is an f32 stream")
    bad(lambda h: (h.connect("w.output", "out"), h.connect("osc.output", "out")), "mixes Frame<2> and f32 sources")
    # a Frame<2> straight to the graph output: a stereo mix bus (two tiles, the ordinary kernel only)
    st = oscen_amd.Graph("st")
    st.output_stream("out")
    st.node("osc", "Oscillator::sine", 330.0, 1.0)
    st.node("w", "Wide::new")
    st.connect("osc.output", "w.input")
    st.connect("w.output", "out")
    src = st.kernel_source()
    assert "og::BusLdsN<2> bus;" in src and "-> og::OutN<2> {" in src and "og_k2_" not in src
    assert st.jit_check() > 0
    # the DSL's typed output (`output out: stream: Frame<2>;`, examples/electric-piano/src/main.rs:51) is kept and checked
    typed = st.to_dsl().replace("output out: stream;", "output out: stream: Frame<2>;")
    g2 = oscen_amd.Graph(dsl=typed)
    assert "output out: stream: Frame<2>;" in g2.to_dsl() and "og::BusLdsN<2> bus;" in g2.kernel_source()
    with pytest.raises(oscen_amd.OscenError, match="declared Frame<2> but fed an f32 stream"):
        oscen_amd.Graph(dsl=typed.replace("w.output -> out", "osc.output -> out")).kernel_source()
    with pytest.raises(oscen_amd.OscenError, match="declared f32 but fed a Frame<2>"):
        oscen_amd.Graph(dsl=typed.replace("stream: Frame<2>", "stream: f32")).kernel_source()
    bad(lambda h: (h.connect("w.output * w.output", "n.input"), h.connect("n.output", "out")), "frame \\* f32")
    oscen_amd.unregister_node("Wide::new")
    oscen_amd.unregister_node("Narrow::new")
