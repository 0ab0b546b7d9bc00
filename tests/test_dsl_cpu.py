"""Text front end for `graph! { ... }` bodies (oscen-graph-compiler/src/parse.rs grammar). CPU only."""
import re
import pytest

import oscen_amd

SYNTH = """
graph! {
    name: Synth;

    // Control inputs with defaults
    input carrier_freq: value = 440.0;
    input mod_depth: value = 0.2 [0.0..1.0, ramp: 1_200];
    input cutoff: value = 1_200.0 {range: 20.0..20000.0, unit: " Hz"};
    input gate: event;
    output audio_out: stream;

    nodes {
        modulator = PolyBlepOscillator::sine(5.0, 0.2);
        carrier = oscen::PolyBlepOscillator::saw(440.0_f32, 0.5);
        filter = TptFilter::new(1200.0, 0.707);
        env = AdsrEnvelope::new(0.01, 0.1, 0.7, 0.2);
    }

    connections {
        carrier_freq -> carrier.frequency;
        mod_depth -> modulator.amplitude;
        cutoff -> filter.cutoff;
        gate -> env.gate;
        modulator.output -> carrier.frequency_mod;
        carrier.output -> filter.input;
        filter.output * env.output -> audio_out;
    }
}
"""


def _builder_synth():
    g = oscen_amd.Graph("Synth")
    g.input_value("carrier_freq", 440.0, per_voice=True)
    g.input_value("mod_depth", 0.2, ramp=1200)
    g.input_value("cutoff", 1200.0)
    g.input_event("gate")
    g.output_stream("audio_out")
    g.node("modulator", "PolyBlepOscillator::sine", 5.0, 0.2)
    g.node("carrier", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("filter", "TptFilter::new", 1200.0, 0.707)
    g.node("env", "AdsrEnvelope::new", 0.01, 0.1, 0.7, 0.2)
    for s, d in [("carrier_freq", "carrier.frequency"), ("mod_depth", "modulator.amplitude"), ("cutoff", "filter.cutoff"),
                 ("gate", "env.gate"), ("modulator.output", "carrier.frequency_mod"), ("carrier.output", "filter.input"),
                 ("filter.output * env.output", "audio_out")]:
        g.connect(s, d)
    return g


def test_dsl_text_lowers_to_the_same_kernel_as_the_builder():
    parsed = oscen_amd.Graph(dsl=SYNTH, per_voice=["carrier_freq"])
    a = parsed.kernel_source()
    b = _builder_synth().kernel_source()
    assert a == b
    assert "input mod_depth: value = 0.200000003 [ramp: 1200];" in parsed.to_dsl()
    assert "input carrier_freq: value = 440.0;  // per voice" in parsed.to_dsl()


@pytest.mark.parametrize("name", ["fm_voice", "sub_voice", "sat4x_voice", "sat1x_voice"])
def test_builtin_graphs_roundtrip_through_the_dsl(name):
    g = oscen_amd.Graph(builtin=name)
    text = g.to_dsl()
    assert text.startswith("name: " + name)
    g2 = oscen_amd.Graph(dsl=text, per_voice=["frequency"])
    assert g2.kernel_source() == g.kernel_source()


def test_old_style_syntax_policies_and_rates():
    text = """
    name: ClipOversampled;
    input value frequency = 220.0;
    output stream out;
    node {
        osc = PolyBlepOscillator::sine(220.0, 0.9);
        clip = HardClip::new() * 4;
    }
    connection {
        frequency -> osc.frequency();
        [sinc_iir] osc.output() -> clip.input();
        [sinc_iir] clip.output() -> out;
    }
    """
    src = oscen_amd.Graph(dsl=text, per_voice=["frequency"]).kernel_source()
    assert "og::iir_up<4>" in src and "og::iir_down<4>" in src


def test_dsl_diagnostics():
    # node arrays parse (round 2); an element type nobody registered is diagnosed when the graph is lowered
    with pytest.raises(oscen_amd.OscenError) as e:
        oscen_amd.Graph(dsl="name: X; nodes { voices = [NobodysVoice::new(); 8]; }").kernel_source()
    assert "unknown node type 'NobodysVoice::new'" in str(e.value)
    with pytest.raises(oscen_amd.OscenError) as e:
        oscen_amd.Graph(dsl="name: X;\ninput a: value = 1.0;\nbogus;")
    assert "line 3" in str(e.value)
    with pytest.raises(oscen_amd.OscenError):
        oscen_amd.Graph(dsl="name: X; input a: value = 1.0;", per_voice=["nope"])
    # a cycle that is not routed through a delay: the reference's "non-feedback cycle" diagnostic (ir/lower.rs:1080)
    with pytest.raises(oscen_amd.OscenError) as e:
        oscen_amd.Graph(dsl="name: X; output o: stream; nodes { a = Gain::new(1.0); b = Gain::new(1.0); } "
                            "connections { a.output -> b.input; b.output -> a.input; b.output -> o; }").kernel_source()
    assert "non-feedback cycle" in str(e.value)
    # only Delay implements AllowsFeedback (oscen-macros/tests/ui/feedback_marker_missing.rs)
    with pytest.raises(oscen_amd.OscenError) as e:
        oscen_amd.Graph(dsl="name: X; output o: stream; nodes { a = Gain::new(1.0); b = Gain::new(1.0); } "
                            "connections { a.output -> [b] -> a.input; a.output -> o; }").kernel_source()
    assert "AllowsFeedback" in str(e.value)
    # `<name>__<digits>` is the spelling of an expanded array element (the endpoint-kind inference shares kinds per array
    # root by that suffix): a user node spelled like one is refused instead of silently sharing kinds with `osc`
    with pytest.raises(oscen_amd.OscenError) as e:
        oscen_amd.Graph(dsl="name: X; output o: stream; nodes { osc = Gain::new(1.0); osc__1 = Gain::new(1.0); } "
                            "connections { osc.output -> osc__1.input; osc__1.output -> o; }").kernel_source()
    assert "reserved for the elements of node arrays" in str(e.value)


def test_dsl_feedback_through_delay():
    """`src -> [N] -> dst` and `src -> [node] -> dst` (ir/lower.rs:342-347): two edges, the second a feedback edge."""
    text = """
    name: Echo;
    input frequency: value = 220.0;
    input gate: event;
    output out: stream;
    nodes {
        osc = PolyBlepOscillator::saw(220.0, 0.5);
        env = AdsrEnvelope::new(0.01, 0.1, 0.5, 0.2);
        mix = Mixer::new();
        fbk = Gain::new(0.4);
        d = Delay::new(100.0, 0.0);
    }
    connections {
        frequency -> osc.frequency;
        gate -> env.gate;
        osc.output * env.output -> mix.input_a;
        mix.output -> [d] -> fbk.input;
        fbk.output -> mix.input_b;
        mix.output -> [3] -> out;
    }
    """
    g = oscen_amd.Graph(dsl=text, per_voice=["frequency"])
    src = g.kernel_source()
    assert src.count("og::delay_tick(") == 2 and "A.rings[1]" in src
    assert "_output_z" in src  # fbk runs before d: it reads d's previous output
    back = g.to_dsl()
    assert "mix.output -> [d] -> fbk.input;" in back and "mix.output -> [3] -> out;" in back
    g2 = oscen_amd.Graph(dsl=back, per_voice=["frequency"])
    assert g2.kernel_source() == src


def test_poly_wrapper_lowering_with_a_registered_voice_type_and_its_diagnostics():
    """MidiParser -> VoiceAllocator<N> -> [MidiVoiceHandler; N] -> [Voice; N] -> sum (examples/fm-synth/src/lib.rs:68-131)
    with a voice graph of one's own: registered as a graph type, the wrapper text lowers to a voice-bank kernel whose
    inputs are the handler-fed per-voice inputs followed by the wrapper's (ramped) parameters; malformed wrappers are
    diagnosed."""
    voice = oscen_amd.Graph(dsl="""
        name: MyVoice;
        input freq: value = 220.0;
        input trig: event;
        input tone: value = 1500.0;
        output audio: stream;
        nodes {
            osc = PolyBlepOscillator::saw(220.0, 0.4);
            env = AdsrEnvelope::new(0.01, 0.1, 0.6, 0.2);
            lp = TptFilter::new(1500.0, 0.8);
        }
        connections {
            freq -> osc.frequency; trig -> env.gate; tone -> lp.cutoff;
            osc.output * env.output -> lp.input; lp.output -> audio;
        }
    """)
    wrapper = """
        name: MyPoly;
        input midi_in: event;
        input brightness: value = 1800.0 [200.0..8000.0, ramp: 441];
        output mix: stream;
        nodes {
            parser = MidiParser::new();
            alloc = VoiceAllocator::<6>::new();
            handlers = [MidiVoiceHandler::new(); 6];
            voices = [MyVoice::new(); 6];
        }
        connections {
            midi_in -> parser.midi_in;
            parser.note_on -> alloc.note_on;  parser.note_off -> alloc.note_off;
            alloc.voices -> handlers.note_on;  alloc.voices -> handlers.note_off;
            handlers.frequency -> voices.freq;  handlers.gate -> voices.trig;
            brightness -> voices.tone;
            voices.audio -> mix;
        }
    """
    with pytest.raises(oscen_amd.OscenError, match="not a graph type"):
        oscen_amd.Graph(dsl=wrapper).kernel_source()
    oscen_amd.register_graph_type("MyVoice", voice)
    try:
        g = oscen_amd.Graph(dsl=wrapper)
        assert g.poly_info() == {"declared_voices": 6, "frequency_input": "freq", "gate_input": "trig"}
        src = g.kernel_source()
        # per-voice frequency input 0, gate event, then the wrapper's ramped parameter (ramp row 0); the voice's own node names
        assert "vin_0" in src and "og::adsr_gate(" in src and "RV(0, " in src
        order = re.search(r"// Node order: (.*)", src).group(1).split()
        assert order == ["osc", "env", "lp"], order
        assert g.jit_check() > 0
        # the same bank written by hand
        flat = oscen_amd.Graph("flat")
        flat.input_value("freq", 220.0, per_voice=True)
        flat.input_event("trig")
        flat.input_value("brightness", 1800.0, ramp=441)
        flat.output_stream("mix")
        flat.node("osc", "PolyBlepOscillator::saw", 220.0, 0.4)
        flat.node("env", "AdsrEnvelope::new", 0.01, 0.1, 0.6, 0.2)
        flat.node("lp", "TptFilter::new", 1500.0, 0.8)
        for a, b in (("freq", "osc.frequency"), ("trig", "env.gate"), ("brightness", "lp.cutoff"),
                     ("osc.output * env.output", "lp.input"), ("lp.output", "mix")):
            flat.connect(a, b)
        strip = lambda t: re.sub(r"from graph '[^']*'|\"(MyPoly|flat)\"", "", t)
        assert strip(flat.kernel_source()) == strip(src)
        # diagnostics
        bad = wrapper.replace("handlers.gate -> voices.trig;", "")
        with pytest.raises(oscen_amd.OscenError, match="gate"):
            oscen_amd.Graph(dsl=bad).kernel_source()
        bad = wrapper.replace("handlers = [MidiVoiceHandler::new(); 6];", "handlers = [MidiVoiceHandler::new(); 4];")
        with pytest.raises(oscen_amd.OscenError, match="4 voice handlers for 6 voices"):
            oscen_amd.Graph(dsl=bad).kernel_source()
        bad = wrapper.replace("voices.audio -> mix;", "voices.audio -> mix; midi_in -> voices.trig;")
        with pytest.raises(oscen_amd.OscenError, match="may only feed the MidiParser"):
            oscen_amd.Graph(dsl=bad).kernel_source()
    finally:
        oscen_amd.unregister_graph_type("MyVoice")


def test_calls_on_a_connection_parse_and_report_like_the_reference():
    """ast.rs:113-129 / parse.rs:1027-1123: `path(args)` calls, `.method(args)`, `[k]` channel index; the diagnostics of
    oscen-macros/tests/ui/{turbofish_non_frame,unknown_node_in_call_arg}.rs"""
    oscen_amd.register_node("DxConst::new", inputs=[], outputs=["output"], n_ctor_args=1, state=[("val", "f32", 0.0, 0)],
                            process="    output = val;\n")
    oscen_amd.register_node("DxStereo::new", inputs=[], outputs=[("output", 2)], process="    output.v[0] = 0.5f;\n    output.v[1] = 0.25f;\n")
    oscen_amd.register_function("dsp::halve", ["x"], "return x * 0.5f;")
    oscen_amd.register_function("swap", [("v", 2)], "og::Frame<2> o; o.v[0] = v.v[1]; o.v[1] = v.v[0]; return o;", result_channels=2)

    def lower(conn, out="stream"):
        g = oscen_amd.Graph(dsl=f"name: Dx; output out: {out}; nodes {{ a = DxConst::new(0.8); s = DxStereo::new(); osc = DxConst::new(0.1); }} "
                                f"connections {{ {conn} }}")
        return g.kernel_source()

    try:
        src = lower("halve(a.output) -> out;")                     # last path segment, like `use dsp::halve;`
        assert "og_fn_dsp__halve(" in src and "return x * 0.5f;" in src
        assert "og_fn_dsp__halve(" in lower("dsp::halve(a.output).abs() + 1.0 -> out;")
        assert "og_fn_swap(" in lower("swap(s.output) -> out;", "stream: Frame<2>")
        assert "og_fn_swap(" in lower("swap(swap(s.output))[0] -> out;")
        assert "tanhf(" in lower("a.output.tanh() -> out;")
        for conn, out, msg in [
            ("dsp::split::<4>(osc.output) -> out;", "stream", "turbofish arguments are only supported on the `Frame` constructor"),
            ("double(ocs.output) -> out;", "stream", "unknown node 'ocs'"),
            ("double(osc.output) -> out;", "stream", "unknown function 'double'"),
            ("halve(s.output) -> out;", "stream", "argument 'x' is an f32, the source is a Frame<2>"),
            ("swap(a.output) -> out;", "stream: Frame<2>", "argument 'v' is a Frame<2>, the source is an f32"),
            ("halve(a.output, a.output) -> out;", "stream", "takes 1 argument(s), 2 given"),
            ("a.output.frobnicate() -> out;", "stream", "unknown f32 method '.frobnicate()'"),
            ("a.output.clamp(0.0) -> out;", "stream", "'.clamp()' takes 2 argument(s), 1 given"),
            ("s.output.tanh() -> out;", "stream", "is an f32 method"),
            ("a.output[0] -> out;", "stream", "needs a Frame<N> source"),
            ("s.output[2] -> out;", "stream", "channel 2 of a Frame<2>"),
            ("dsp::a.output -> out;", "stream", "a path is only valid as a function name"),
            ("Frame(a.output) -> out;", "stream: Frame<2>", "Frame(..) takes 2 to 4 channels"),
            ("Frame(s.output, a.output) -> out;", "stream: Frame<2>", "Frame(..) takes f32 channels"),
        ]:
            with pytest.raises(oscen_amd.OscenError) as ei:
                lower(conn, out)
            assert msg in str(ei.value), (conn, str(ei.value))
        for bad in ("", "1x", "a::", "Frame", "x::Frame"):
            with pytest.raises(oscen_amd.OscenError):
                oscen_amd.register_function(bad, ["x"], "return x;")
        with pytest.raises(oscen_amd.OscenError):
            oscen_amd.register_function("f", [("x", 9)], "return x;")
    finally:
        oscen_amd.unregister_node("DxConst::new")
        oscen_amd.unregister_node("DxStereo::new")
        oscen_amd.unregister_function("dsp::halve")
        oscen_amd.unregister_function("swap")
    with pytest.raises(oscen_amd.OscenError):
        oscen_amd.unregister_function("swap")


def test_calls_on_a_connection_compile_in_every_kernel_shape():
    """named functions, methods and frame constructors inside the shapes the generator treats differently -- across a
    rate boundary, in array broadcasts, inside a nested graph, feeding a Frame<2> output of a graph whose nodes would
    also get the pipelined kernels (round 4: the last wave keeps one bus tile per channel) -- each compiled
    for gfx950 (og_graph_jit_check)"""
    oscen_amd.register_function("half", ["x"], "return x * 0.5f;")
    oscen_amd.register_function("ms", [("v", 2)], "og::Frame<2> o; o.v[0] = v.v[0] - v.v[1]; o.v[1] = v.v[0] + v.v[1]; return o;",
                                result_channels=2)
    inner = oscen_amd.Graph(dsl="name: DxInnerFn; input x: stream; output y: stream; nodes { c = HardClip::new(); } "
                                "connections { half(x) * 3.0 -> c.input; half(c.output).abs() -> y; }")
    oscen_amd.register_graph_type("DxInnerFn", inner)
    try:
        cases = [
            ("a = PolyBlepOscillator::saw(220.0, 0.5) * 2;", "[sinc] half(a.output) -> out;", "stream"),
            ("a = PolyBlepOscillator::saw(220.0, 0.5); c = HardClip::new() * 4;", "[linear] half(a.output).tanh() -> c.input; [sinc] c.output -> out;", "stream"),
            ("oscs = [PolyBlepOscillator::saw(220.0, 0.5); 3];", "half(oscs[1].output) + oscs[2].output.abs() -> out;", "stream"),
            ("oscs = [PolyBlepOscillator::saw(220.0, 0.5); 2]; g = [Gain::new(0.5); 2];", "half(oscs.output) -> g.input; g.output -> out;", "stream"),
            ("a = PolyBlepOscillator::saw(220.0, 0.5); n = DxInnerFn::new();", "a.output -> n.x; n.y -> out;", "stream"),
            ("a = PolyBlepOscillator::saw(220.0, 0.5); b = PolyBlepOscillator::sine(330.0, 0.5);", "ms(Frame(a.output, b.output)) * 0.5 -> out;", "stream: Frame<2>"),
            ("a = PolyBlepOscillator::saw(220.0, 0.5); f = TptFilter::new(1000.0, 0.7);",
             "a.output -> f.input; cutoff.clamp(100.0, 4000.0) + half(a.output) * 100.0 -> f.cutoff; f.output -> out;", "stream"),
        ]
        for nodes, conns, ty in cases:
            g = oscen_amd.Graph(dsl=f"name: DxT; input cutoff: value = 800.0; output out: {ty}; nodes {{ {nodes} }} connections {{ {conns} }}")
            src = g.kernel_source()
            assert "og_fn_half(" in src or "og_fn_ms(" in src
            if ty != "stream":  # round 4: a frame-valued output runs in the pipelined kernels too (N bus tiles in the last wave)
                assert "voice_block_p2" in src and "og::BusLdsN<2> bus" in src[src.index("voice_block_p2"):]
            assert g.jit_check() > 1000, conns
    finally:
        oscen_amd.unregister_graph_type("DxInnerFn")
        oscen_amd.unregister_function("half")
        oscen_amd.unregister_function("ms")


def test_policy_fed_nested_input_read_inside_a_compound_expression():
    """`[linear] a.output -> n.x` into an oversampled nested graph whose body reads `x` inside compound expressions
    (`half(x) * 3.0 -> c.input`): the outer source is resampled ONCE into the inner input field -- a unit-gain node of the
    nested rate -- and the expressions read the field: the same kernel as the hand-flattened graph"""
    oscen_amd.register_function("half", ["x"], "return x * 0.5f;")
    inner = oscen_amd.Graph(dsl="name: DxPInner; input x: stream; output y: stream; nodes { c = HardClip::new(); } "
                                "connections { half(x) * 3.0 -> c.input; half(c.output).abs() + x * 0.1 -> y; }")
    oscen_amd.register_graph_type("DxPInner", inner)
    try:
        nested = oscen_amd.Graph(dsl="""name: DxPO; input frequency: value = 220.0; output out: stream;
            nodes { a = PolyBlepOscillator::saw(220.0, 0.5); n = DxPInner::new() * 2; }
            connections { frequency -> a.frequency; [linear] a.output -> n.x; [sinc] n.y -> out; }""", per_voice=["frequency"])
        flat = oscen_amd.Graph(dsl="""name: DxPO; input frequency: value = 220.0; output out: stream;
            nodes { a = PolyBlepOscillator::saw(220.0, 0.5); n_c = HardClip::new() * 2; n_in__x = Gain::new(1.0) * 2; }
            connections { frequency -> a.frequency; [linear] a.output -> n_in__x.input; half(n_in__x.output) * 3.0 -> n_c.input;
                          [sinc] half(n_c.output).abs() + n_in__x.output * 0.1 -> out; }""", per_voice=["frequency"])
        assert nested.kernel_source() == flat.kernel_source()
        assert nested.jit_check() > 1000
    finally:
        oscen_amd.unregister_graph_type("DxPInner")
        oscen_amd.unregister_function("half")


def test_several_edges_into_a_value_destination_follow_the_reference_kind_propagation():
    """ir/lower.rs:233-338 + codegen/emit_node.rs:35-58: typed graph endpoints and stream-only policies seed endpoint kinds,
    connection statements propagate them both ways; several edges into a destination KNOWN to be a value are assigned one
    after the other (the last wins), several edges into a stream or an untyped destination are summed"""
    head = "name: DxK; input cutoff: value = 900.0; input frequency: value = 220.0; output out: stream; "
    nodes = "nodes { osc = PolyBlepOscillator::saw(220.0, 0.5); lfo = PolyBlepOscillator::sine(3.0, 200.0); f = TptFilter::new(1000.0, 0.7); } "

    def src(conns):
        g = oscen_amd.Graph(dsl=head + nodes + "connections { frequency -> osc.frequency; osc.output -> f.input; f.output -> out; " + conns + " }",
                            per_voice=["frequency"])
        return g.kernel_source()

    only_lfo = src("lfo.output -> f.cutoff;")
    only_cut = src("cutoff -> f.cutoff;")
    summed = src("cutoff + lfo.output -> f.cutoff;")
    assert len({only_lfo, only_cut, summed}) == 3
    # `cutoff` is a typed value input: f.cutoff is a value endpoint, the later edge replaces the earlier one
    assert src("cutoff -> f.cutoff; lfo.output -> f.cutoff;") == only_lfo
    assert src("lfo.output -> f.cutoff; cutoff -> f.cutoff;") == only_cut
    # a literal-scaled value input is still a value (Binary(Value, Value)); a call has no inferred kind, so nothing types
    # the destination and the edges are summed
    assert src("cutoff * 0.5 -> f.cutoff; lfo.output -> f.cutoff;") == only_lfo
    # untyped node-to-node endpoints: the sum, in edge order
    two = src("lfo.output -> f.cutoff; osc.output -> f.cutoff;")
    assert "(n1_output + " in two or "+ n0_output" in two or " + " in two
    assert two != only_lfo
    # `osc.output -> out` types osc.output as a STREAM (out is a typed stream output), which types f.cutoff when osc feeds
    # it first: a stream destination sums even when a value input joins later
    def src2(conns):  # ... with osc.output also wired to a typed stream output
        g = oscen_amd.Graph(dsl="name: DxK; input cutoff: value = 900.0; input frequency: value = 220.0; output out: stream; output raw: stream; "
                                + nodes + "connections { frequency -> osc.frequency; osc.output -> f.input; f.output -> out; osc.output -> raw; "
                                + conns + " }", per_voice=["frequency"])
        return g.kernel_source()

    assert src("osc.output -> f.cutoff; cutoff -> f.cutoff;") == only_cut          # osc.output untyped here: the value input wins
    s1 = src2("osc.output -> f.cutoff; cutoff -> f.cutoff;")
    assert s1 not in (src2("cutoff -> f.cutoff;"), src2("osc.output -> f.cutoff;"))  # typed stream: the sum
    # a stream-only policy seeds the kind as well (both ends of a [linear] / [sinc] / [sinc_iir] edge are streams): c.output
    # is a stream, so g.gain is a stream destination and `amount` joins the sum instead of replacing it
    def src3(gain_conns):
        g = oscen_amd.Graph(dsl="""name: DxK2; input amount: value = 0.5; output out: stream; output aux: stream;
            nodes { a = PolyBlepOscillator::saw(220.0, 0.5); c = HardClip::new(); g = Gain::new(1.0); up = HardClip::new() * 2; }
            connections { a.output -> c.input; [linear] c.output -> up.input; [sinc] up.output -> aux; a.output -> g.input; g.output -> out; """
                            + gain_conns + " }")
        return g.kernel_source()

    assert src3("c.output -> g.gain; amount -> g.gain;") == src3("c.output + amount -> g.gain;")
    # ... but only same-rate sources can be summed: the same fan-in into an OVERSAMPLED node takes `amount` across a rate
    # boundary, which the reference refuses (codegen/emit_node.rs:87-89 "a cross-rate edge") -- and so does this compiler
    with pytest.raises(oscen_amd.OscenError, match="fan-in summing supports only same-rate scalar/frame stream sources; saw a cross-rate edge into `g.gain`"):
        oscen_amd.Graph(dsl="""name: DxK2x; input amount: value = 0.5; output out: stream; output aux: stream;
            nodes { a = PolyBlepOscillator::saw(220.0, 0.5) * 2; c = HardClip::new() * 2; g = Gain::new(1.0) * 2; }
            connections { a.output -> c.input; [sinc] c.output -> aux; a.output -> g.input; [sinc] g.output -> out;
                          c.output -> g.gain; amount -> g.gain; }""").kernel_source()
    # one kind per (ROOT node, field): a typed value input wired to ONE element of an array makes the port a value endpoint
    # of every element (the reference infers kinds before it unrolls the array) -- two edges into another element: last wins
    def src4(conns):
        g = oscen_amd.Graph(dsl="""name: DxK3; input cutoff: value = 900.0; output out: stream;
            nodes { osc = PolyBlepOscillator::saw(220.0, 0.5); lfo = PolyBlepOscillator::sine(3.0, 200.0); lfo2 = PolyBlepOscillator::sine(5.0, 100.0);
                    fs = [TptFilter::new(1000.0, 0.7); 2]; }
            connections { osc.output -> fs.input; fs.output -> out; cutoff -> fs[0].cutoff; """ + conns + " }")
        return g.kernel_source()

    assert src4("lfo.output -> fs[1].cutoff; lfo2.output -> fs[1].cutoff;") == src4("lfo2.output -> fs[1].cutoff;")


def test_round4_front_end_edges_follow_the_reference():
    """(1) a connection policy on a SAME-rate edge is accepted and ignored: classify_edge_ir `(Same, Same) =>
    EdgeKernel::None` whatever the policy (oscen-graph-compiler/src/ir/lower.rs:870-880) -- the `_1x` expansion of
    oversample_variants! keeps its `[sinc]` edge (oscen-lib/tests/oversample_variants.rs:8-23);
    (2) a `[ramp: N]` input of a NESTED graph is a ValueRampState the nested process() ticks itself
    (codegen/mod.rs:559-572): nothing outside can move it, it idles at its default; an outer edge into it is an error
    (no ConnectEndpoints<f32, ValueRampState> impl, graph/static_context.rs:41-147);
    (3) `frame_offset` is a parameter of the generated handlers: reserved as a port / field name."""
    plain = oscen_amd.Graph(dsl="name: P; output stream out; nodes { osc = PolyBlepOscillator::saw(440.0, 0.6); } connections { osc.output -> out; }")
    for pol in ("sinc", "linear", "latch", "sinc_iir"):
        g = oscen_amd.Graph(dsl="name: P; output stream out; nodes { osc = PolyBlepOscillator::saw(440.0, 0.6); } connections { [%s] osc.output -> out; }" % pol)
        assert g.kernel_source() == plain.kernel_source()
    inner_ramped = oscen_amd.Graph(dsl="name: R4Inner; input value level = 0.25 [ramp: 64]; output stream out; "
                                       "nodes { osc = PolyBlepOscillator::sine(330.0, 1.0); } connections { osc.output * level -> out; }")
    inner_plain = oscen_amd.Graph(dsl="name: R4InnerP; input value level = 0.25; output stream out; "
                                      "nodes { osc = PolyBlepOscillator::sine(330.0, 1.0); } connections { osc.output * level -> out; }")
    oscen_amd.register_graph_type("R4Inner", inner_ramped)
    oscen_amd.register_graph_type("R4InnerP", inner_plain)
    try:
        a = oscen_amd.Graph(dsl="name: O; output stream out; nodes { v = R4Inner; } connections { v.out -> out; }")
        b = oscen_amd.Graph(dsl="name: O; output stream out; nodes { v = R4InnerP; } connections { v.out -> out; }")
        assert a.kernel_source() == b.kernel_source()  # the idle ramp is the constant 0.25
        with pytest.raises(oscen_amd.OscenError, match="ramped input"):
            oscen_amd.Graph(dsl="name: O2; input value x = 0.5; output stream out; nodes { v = R4Inner; } "
                                "connections { x -> v.level; v.out -> out; }").kernel_source()
    finally:
        oscen_amd.unregister_graph_type("R4Inner")
        oscen_amd.unregister_graph_type("R4InnerP")
    with pytest.raises(oscen_amd.OscenError, match="reserved"):
        oscen_amd.register_node("R4Bad::new", inputs=[("frame_offset", "value", 0.0, -1)], outputs=["out"], process="    out = frame_offset;\n")
