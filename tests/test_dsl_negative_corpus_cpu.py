"""The reference's COMPILE-FAIL corpus (oscen-macros/tests/ui/*.rs, trybuild fixtures: every `graph!` body there is wrong on
purpose, the `.stderr` beside it holds what rustc prints) against this front end: every body must be refused by
og_graph_parse / the lowering, and the diagnostic must name the same construct as the `.stderr`.

Nothing is copied: the fixtures are read where they lie and the test skips when the checkout is absent (the GPU box).

How "the same construct" is checked, per fixture:
  * a phrase that must appear in BOTH the `.stderr` and this library's message (the rule's own wording wherever this
    library's diagnostic restates it -- ir/lower.rs:459-490 kind mismatch, codegen/emit_node.rs:35-125 fan-in, parse.rs
    rate rules), or a pair (phrase of the `.stderr`, phrase of ours) where the wording differs but the rule is the same;
  * for parse errors, the LINE: this library counts lines from the start of the body, the `.stderr` from the start of
    the file -- translated, both must point at the same source line.
Node types a fixture declares locally in Rust are registered as stubs with the struct's ports (as test_dsl_corpus_cpu.py
does); a local type that SHADOWS a type this library ships (`Delay` in feedback_marker_missing.rs: the C ABI has one
type namespace, Rust has paths) is registered under `Local<Type>` and the body's constructor renamed accordingly.
"""
import os
import re

import pytest

import oscen_amd
from tests.test_dsl_corpus_cpu import REF, register_stub, rust_node_structs, strip_comments

UI = os.path.join(REF, "oscen-macros", "tests", "ui")

# fixture -> (phrase in the .stderr, phrase in our diagnostic, check the line?)
EXPECT = {
    "array_with_double_rate.rs": ("node already has an embedded rate (`* N` or `/ N`) from the array literal", None, True),
    "array_with_invalid_rate.rs": ("rate factor must be 1, 2, 4, or 8", None, False),
    "cross_rate_kind_mismatch.rs": ("no connection from EventOutput to f32", "event source 'ee.gate' cannot feed 'ss.input'", False),
    "down_rate.rs": ("node undersampling (`/ N`) is not", None, True),
    "fanin_unsupported.rs": ("fan-in summing supports only same-rate scalar/frame stream sources; saw a compound (non-endpoint) source into `filter.input`",
                             None, False),
    "feedback_marker_missing.rs": ("AllowsFeedback", None, False),
    "frame_turbofish_no_args.rs": ("expected `(` after `Frame::<N>`", None, False),
    "invalid_rate.rs": ("rate factor must be 1, 2, 4, or 8", None, False),
    "mixed_rates.rs": ("v1 does not support connections between two differently-rated non-default-rate nodes", None, False),
    # two errors in the reference (it collects them); this front end stops at the first one, the same as the reference's first
    "multi_error_mixed.rs": ("node undersampling (`/ N`) is not", None, True),
    "multi_error_type_mismatch.rs": ("Type mismatch in connection: source is Stream but destination expects Value", None, False),
    "multi_parse_error_in_connection_block.rs": ("expected `-`", "expected `->` in connection", True),
    "multi_parse_error_in_node_block.rs": ("expected `;`", "expected ';'", True),
    "multi_parse_error_top_level.rs": ("expected `;`", "expected ';'", True),
    "turbofish_non_frame.rs": ("turbofish arguments are only supported on the `Frame` constructor", None, False),
    "unknown_node_in_call_arg.rs": ("unknown node or endpoint name", "unknown node 'ocs'", False),
}

# refused as well, but for a reason of this library's scope rather than by the fixture's rule -- each with its reason
ALLOW = {
    "asset_endpoint_mismatch.rs": "`external` asset handles (sample players / convolvers) are out of scope (SURVEY 2): the body is refused at "
                                  "its `external` declaration, before the stream -> asset edge the fixture is about",
    "two_sample_rate_fields.rs": "a #[derive(Node)] rule (at most one SampleRate field), no graph body: node types reach this library through "
                                 "og_register_node, whose bodies see ONE implicit `sample_rate` and cannot declare such fields",
}

SHIPPED = ("AdsrEnvelope", "PolyBlepOscillator", "Oscillator", "TptFilter", "Gain", "FmOperator", "Crossfade", "Mixer", "AddValue", "Vca",
           "HardClip", "IirLowpass", "Delay", "LP18Filter", "Tremolo")


def fixture(name):
    raw = open(os.path.join(UI, name), encoding="utf-8").read()
    t = strip_comments(raw)
    m = re.search(r"\bgraph!\s*\{", t)
    if not m:
        return raw, t, None, 0
    i, depth = m.end(), 1
    while i < len(t) and depth:
        depth += {"{": 1, "}": -1}.get(t[i], 0)
        i += 1
    return raw, t, t[m.end(): i - 1], t.count("\n", 0, m.end()) + 1  # (the file line the body starts on)


def refuse(body, structs):
    """parse + lower with stub node types for the structs the file declares; returns the diagnostic, or None if a kernel came out"""
    registered = []
    try:
        for ty in structs:
            if ty in SHIPPED:  # a local type shadowing a shipped one: one type namespace behind the C ABI
                body = re.sub(r"\b%s::" % ty, "Local%s::" % ty, body)
        for attempt in range(16):
            try:
                g = oscen_amd.Graph(dsl=body)
                g.kernel_source()
                return None
            except oscen_amd.OscenError as e:
                msg = str(e)
                m = re.search(r"unknown node type '([^']+)'", msg)
                if m and m.group(1) not in registered:
                    base = m.group(1).split("::")[0]
                    key = base[len("Local"):] if base.startswith("Local") and base[len("Local"):] in structs else base
                    if key in structs and register_stub(m.group(1), structs[key], 0) is None:
                        registered.append(m.group(1))
                        continue
                return msg
        return "gave up"
    finally:
        for ty in registered:
            oscen_amd.unregister_node(ty)


@pytest.mark.skipif(not os.path.isdir(UI), reason="reference checkout not present (nothing of it is copied into the repo)")
def test_every_compile_fail_fixture_of_the_reference_is_refused_for_the_reason_the_reference_gives():
    files = sorted(f for f in os.listdir(UI) if f.endswith(".rs"))
    assert len(files) >= 18 and set(files) == set(EXPECT) | set(ALLOW), sorted(set(files) ^ (set(EXPECT) | set(ALLOW)))
    for f in files:
        raw, text, body, body_line = fixture(f)
        stderr = open(os.path.join(UI, f[:-3] + ".stderr"), encoding="utf-8").read()
        if body is None:
            assert f in ALLOW, f
            continue
        msg = refuse(body, rust_node_structs(text))
        assert msg is not None, "%s: the reference refuses this body, this front end lowered it" % f
        if f in ALLOW:
            assert "external" in msg, (f, msg)
            continue
        theirs, ours, check_line = EXPECT[f]
        assert theirs in stderr, (f, theirs)
        assert (ours or theirs) in msg, (f, msg)
        if check_line:
            m = re.search(r"line (\d+):", msg)
            first = re.search(r"--> tests/ui/%s:(\d+):" % re.escape(f), stderr)
            assert m and first, (f, msg)
            assert body_line + int(m.group(1)) - 1 == int(first.group(1)), (f, msg, first.group(0))


@pytest.mark.skipif(not os.path.isdir(UI), reason="reference checkout not present")
def test_the_second_error_of_a_two_error_fixture_is_reported_once_the_first_is_fixed():
    """multi_error_mixed.rs holds two independent mistakes (`/ 2` and `s -> v_out`); the reference reports both at once, this
    front end the first -- and the second as soon as the first is gone"""
    _, text, body, _ = fixture("multi_error_mixed.rs")
    fixed = body.replace("/ 2", "")
    assert fixed != body
    msg = refuse(fixed, {})
    assert msg and "Type mismatch in connection: source is Stream but destination expects Value" in msg, msg


def test_connection_rules_of_the_reference_without_the_checkout():
    """the same rules on bodies written here (they run on the GPU box too, where the checkout is absent)"""
    def err(dsl):
        with pytest.raises(oscen_amd.OscenError) as e:
            oscen_amd.Graph(dsl=dsl).kernel_source()
        return str(e.value)

    head = "name: N1; input amount: value = 0.5; input s: stream; input trig: event; output out: stream; output dry: stream; output v: value; "
    nodes = "nodes { a = PolyBlepOscillator::saw(220.0, 0.5); b = PolyBlepOscillator::saw(330.0, 0.5); f = TptFilter::new(900.0, 0.7); } "
    # kind pairs the reference accepts between TYPED ends: S->S, V->V, V->S (types_compatible, ir/lower.rs:1157-1165) ...
    ok = oscen_amd.Graph(dsl=head + nodes + "connections { a.output -> f.input; s + amount -> dry; f.output -> out; amount -> v; }")
    assert "og_k_" in ok.kernel_source()
    # ... and the ones it refuses
    assert "source is Stream but destination expects Value" in err(head + nodes + "connections { s -> v; a.output -> out; }")
    assert "source is Stream but destination expects Value" in err(head + nodes + "connections { s * amount -> v; a.output -> out; }")
    # kinds spread along statements: `a.output -> out` types a.output as a stream, `amount -> f.cutoff` types f.cutoff as a value
    assert "source is Stream but destination expects Value" in err(
        head + nodes + "connections { a.output -> out; amount -> f.cutoff; a.output -> f.cutoff; a.output -> f.input; f.output -> out; }")
    # summed fan-in: plain endpoints at one rate only
    assert "saw a compound (non-endpoint) source into `f.input`" in err(head + nodes + "connections { a.output * 0.5 -> f.input; b.output -> f.input; f.output -> out; }")
    assert "saw a compound (non-endpoint) source into `out`" in err(head + nodes + "connections { a.output -> out; b.output.tanh() -> out; }")
    x4 = "nodes { a = PolyBlepOscillator::saw(220.0, 0.5) * 4; b = PolyBlepOscillator::saw(330.0, 0.5); c = HardClip::new() * 4; } "
    assert "saw a cross-rate edge into `c.input`" in err("name: N2; output out: stream; " + x4 + "connections { a.output -> c.input; b.output -> c.input; [sinc] c.output -> out; }")
    arr = "nodes { os = [PolyBlepOscillator::saw(220.0, 0.5); 3]; b = PolyBlepOscillator::saw(330.0, 0.5); fs = [TptFilter::new(900.0, 0.7); 3]; f = TptFilter::new(500.0, 0.7); } "
    assert "saw an array (parallel) source into `fs.input`" in err("name: N3; output out: stream; " + arr + "connections { os.output -> fs.input; os.output -> fs.input; fs.output -> out; }")
    assert "saw a broadcast source into `fs.input`" in err("name: N4; output out: stream; " + arr + "connections { b.output -> fs.input; os.output -> fs.input; fs.output -> out; }")
    assert "saw an array fan-in source into `f.input`" in err("name: N5; output out: stream; " + arr + "connections { os.output -> f.input; b.output -> f.input; f.output -> out; }")
    # a destination KNOWN to be a value is not a sum (last write wins): compound sources are fine there
    lw = oscen_amd.Graph(dsl=head + nodes + "connections { amount -> f.cutoff; amount * 2.0 -> f.cutoff; a.output -> f.input; f.output -> out; }")
    assert "og_k_" in lw.kernel_source()
    # `-> [name] ->` needs a node type that implements AllowsFeedback (oscen-lib/src/delay/mod.rs:85: Delay, and nothing else here)
    assert "AllowsFeedback" in err("name: N6; input s: stream; output out: stream; nodes { g = TptFilter::new(1000.0, 0.7); d = Gain::new(1.0); } "
                                   "connections { s -> g.input; g.output -> [d] -> g.input; g.output -> out; }")
    fb = oscen_amd.Graph(dsl="name: N7; input s: stream; output out: stream; nodes { g = TptFilter::new(1000.0, 0.7); d = Delay::new(4.0, 0.0); } "
                             "connections { s -> g.input; g.output -> [d] -> g.input; g.output -> out; }")
    assert "og_k_" in fb.kernel_source()
