"""Compile-only fuzz of the front end (CPU): random small `graph!` bodies mixing what round 3 added -- several stream
outputs, outputs read as sources, Frame<2> values, frame constructors, channel extraction, named functions, f32 methods,
node arrays, oversampled nodes with cross-rate policies -- are lowered and compiled for gfx950 (og_graph_jit_check).
A body may be REJECTED by the lowering with a diagnostic (some combinations are documented as unsupported); what must
never happen is generated source that does not compile, or a crash.  (Numerical fuzzing against an interpreter over the
oracle's nodes is tests/test_codegen_fuzz_gpu.py.)"""
import numpy as np
import pytest

import oscen_amd

N_GRAPHS = 24  # (round 6: the ordinary kernel runs longer straight-line regions: each compile takes half as long again)


def build(rng, k):
    rate_choices = ["", "", "", " * 2", " * 4"]
    nodes, scalars, frames = [], [], []  # (expr, inner_rate_tag)
    n_osc = int(rng.integers(1, 4))
    for i in range(n_osc):
        rate = rate_choices[int(rng.integers(0, len(rate_choices)))]
        kind = ["PolyBlepOscillator::saw", "PolyBlepOscillator::sine", "PolyBlepOscillator::square"][int(rng.integers(0, 3))]
        nodes.append(f"o{i} = {kind}({110.0 * (i + 1):.1f}, 0.4){rate};")
        scalars.append((f"o{i}.output", rate))
    if rng.random() < 0.5:
        nodes.append("st = FzStereo::new(0.3);")
        frames.append(("st.out", ""))
    conns = []
    if frames and scalars:
        src = [s for s in scalars if s[1] == ""]
        if src:
            conns.append(f"{src[0][0]} -> st.inp;")
    if rng.random() < 0.4:
        nodes.append("arr = [Gain::new(0.5); 3];")
        src = [s for s in scalars if s[1] == ""]
        if src:
            conns.append(f"{pick_expr(rng, src, [], 1)} -> arr.input;")
            scalars.append(("arr[1].output", ""))

    outs = []
    n_out = int(rng.integers(1, 4))
    budget = 4
    for oi in range(n_out):
        want_frame = bool(frames or rng.random() < 0.3) and rng.random() < 0.45 and budget >= 2
        width = 2 if want_frame else 1
        if budget < width:
            break
        budget -= width
        outs.append((f"out{oi}", width))
    decls = "".join(f"output {n}: stream{': Frame<2>' if w == 2 else ''}; " for n, w in outs)
    for oi, (name, width) in enumerate(outs):
        # one rate domain per expression: pick the sources of one domain, cross with a policy when it is not the outer one
        dom = scalars[int(rng.integers(0, len(scalars)))][1]
        pool = [s for s in scalars if s[1] == dom]
        fpool = frames if dom == "" else []
        if oi > 0 and rng.random() < 0.3 and dom == "":  # an earlier output as a source
            prev = outs[int(rng.integers(0, oi))]
            (pool if prev[1] == 1 else fpool).append((prev[0], ""))
        expr = pick_expr(rng, pool, fpool, width)
        policy = "" if dom == "" else ["[sinc] ", "[linear] ", "[sinc_iir] ", "[latch] "][int(rng.integers(0, 4))]
        conns.append(f"{policy}{expr} -> {name};")
    body = f"name: Fz{k}; input cutoff: value = 900.0; {decls} nodes {{ {' '.join(nodes)} }} connections {{ {' '.join(conns)} }}"
    return body


def pick_expr(rng, scalars, frames, width, depth=0):
    def scalar(d):
        r = rng.random()
        if d > 2 or r < 0.3 or not scalars:
            if frames and rng.random() < 0.3:
                return f"{frames[int(rng.integers(0, len(frames)))][0]}[{int(rng.integers(0, 2))}]"
            if not scalars:
                return "0.25"
            return scalars[int(rng.integers(0, len(scalars)))][0]
        if r < 0.5:
            return f"({scalar(d + 1)} {'+-*'[int(rng.integers(0, 3))]} {scalar(d + 1)})"
        if r < 0.65:
            m = [".tanh()", ".abs()", ".clamp(-0.5, 0.5)", ".max(0.1)", ".sin()", ".powi(2)"][int(rng.integers(0, 6))]
            return f"{scalar(d + 1)}{m}"
        if r < 0.8:
            return f"fz::half({scalar(d + 1)})"
        if r < 0.9:
            return f"({scalar(d + 1)} * 0.5)"
        return f"swap2({frame(d + 1)})[{int(rng.integers(0, 2))}]"

    def frame(d):
        r = rng.random()
        if frames and (d > 2 or r < 0.3):
            return frames[int(rng.integers(0, len(frames)))][0]
        if r < 0.6 or d > 2:
            return f"Frame::<2>({scalar(d + 1)}, {scalar(d + 1)})"
        if r < 0.8:
            return f"swap2({frame(d + 1)})"
        if r < 0.9:
            return f"({frame(d + 1)} * 0.5)"
        return f"({frame(d + 1)} + {frame(d + 1)})"

    return frame(depth) if width == 2 else scalar(depth)


def test_random_bodies_lower_and_compile_or_are_rejected_with_a_diagnostic():
    oscen_amd.register_node("FzStereo::new", inputs=[("inp", "stream", 0.0, -1), ("pan", "value", 0.5, 0)], outputs=[("out", 2)], n_ctor_args=1,
                            process="    out.v[0] = inp * (1.0f - pan);\n    out.v[1] = inp * pan;\n")
    oscen_amd.register_function("fz::half", ["x"], "return x * 0.5f;")
    oscen_amd.register_function("swap2", [("v", 2)], "og::Frame<2> o; o.v[0] = v.v[1]; o.v[1] = v.v[0]; return o;", result_channels=2)
    compiled, rejected = 0, []
    try:
        rng = np.random.default_rng(20260928)
        for k in range(N_GRAPHS):
            body = build(rng, k)
            try:
                g = oscen_amd.Graph(dsl=body)
                g.kernel_source()
            except oscen_amd.OscenError as e:
                msg = str(e)
                assert "oscen graph" in msg and "internal" not in msg, (body, msg)
                rejected.append((body, msg))
                continue
            try:
                assert g.jit_check() > 500
            except oscen_amd.OscenError as e:
                raise AssertionError("generated source does not compile:\n%s\n%s" % (body, str(e)[:1500]))
            compiled += 1
        print("\n%d compiled, %d rejected: %s" % (compiled, len(rejected), sorted({m.split("oscen graph: ")[-1][:60] for _, m in rejected})))
        assert compiled >= N_GRAPHS * 2 // 3, (compiled, [m for _, m in rejected][:5])
    finally:
        oscen_amd.unregister_node("FzStereo::new")
        oscen_amd.unregister_function("fz::half")
        oscen_amd.unregister_function("swap2")


def test_mutated_bodies_and_garbage_expressions_end_in_a_diagnostic():
    """robustness of the front end: graph! bodies with random insertions / deletions / duplications / truncations and
    connection sources made of random tokens either lower or raise OscenError -- no crash, no hang (a `graph!` body with a
    mistake in it must come back as a message)"""
    import random

    rnd = random.Random(7)
    seeds = [oscen_amd.Graph(builtin=b).to_dsl() for b in ("fm_voice", "sub_voice", "sat4x_voice", "echo_voice")]
    seeds.append("""name: Wrap; input midi_in: event; input cutoff: value = 900.0 [ramp: 64]; output out: stream: Frame<2>;
        nodes { p = MidiParser::new(); va = VoiceAllocator::<4>::new(); h = [MidiVoiceHandler::new(); 4]; voices = [FMVoice::new(); 4]; }
        connections { midi_in -> p.midi_in; p.note_on -> va.note_on; p.note_off -> va.note_off; va.voices -> h.note_on; va.voices -> h.note_off;
                      h.frequency -> voices.frequency; h.gate -> voices.gate; cutoff -> voices.filter_cutoff; voices.audio_out -> out; }""")
    tokens = ["->", ";", "{", "}", "(", ")", "[", "]", "*", "+", "-", "/", ".", ",", "::", "<", ">", "=", "Frame", "0.5", "1e9", "nodes",
              "connections", "input", "output", "stream", "value", "event", "sinc", "x", "a.output", "[a; 3]", "* 2", "::<2>", '"', "'", "\\", "\n"]
    lowered = rejected = 0
    for _ in range(400):
        b = rnd.choice(seeds)
        for _ in range(rnd.randint(1, 4)):
            r, pos = rnd.random(), rnd.randint(0, len(b))
            if r < 0.4:
                b = b[:pos] + rnd.choice(tokens) + b[pos:]
            elif r < 0.7:
                b = b[:pos] + b[min(len(b), pos + rnd.randint(1, 12)):]
            elif r < 0.85:
                e = min(len(b), pos + rnd.randint(1, 30))
                b = b[:pos] + b[pos:e] * 2 + b[e:]
            else:
                b = b[:pos]
        try:
            oscen_amd.Graph(dsl=b).kernel_source()
            lowered += 1
        except oscen_amd.OscenError:
            rejected += 1
    atoms = ["a.output", "b.output", "cutoff", "0.5", "2", "(", ")", "+", "-", "*", "/", ".", ",", "[0]", "[9]", "Frame", "Frame::<2>", "::", "half",
             ".tanh()", ".clamp(0.0, 1.0)", ".max(", ".abs", "out0", " ", "nobody.output", "a.nothing", "a", "[", "]", "<", ">"]
    for _ in range(600):
        expr = "".join(rnd.choice(atoms) + rnd.choice(["", " "]) for _ in range(rnd.randint(1, 9)))
        g = oscen_amd.Graph("ef")
        g.input_value("cutoff", 900.0)
        g.output_stream("out0")
        g.output_stream("out1")
        g.node("a", "PolyBlepOscillator::saw", 220.0, 0.5)
        g.node("b", "PolyBlepOscillator::sine", 330.0, 0.5)
        try:
            g.connect("b.output", "out0")
            g.connect(expr, "out1")
            g.kernel_source()
            lowered += 1
        except oscen_amd.OscenError:
            rejected += 1
    assert lowered > 10 and rejected > 500, (lowered, rejected)
