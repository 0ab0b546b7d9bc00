"""Random voice graphs end to end: graph compiler -> hiprtc -> MI355X, against a per-sample interpreter of
the same description over the oracle's nodes (tests/graph_interp.py).  Checks the COMPILER (schedule,
fan-in sums, compound sources, feedback edges, hoisting, ramp tables, pipelined variants) on graphs
nobody hand-expanded."""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests.graph_interp import VoiceInterp
from tests.test_codegen_fuzz_cpu import random_graph

pytestmark = pytest.mark.gpu
SR = 48000.0


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_random_graph_matches_the_node_interpreter(seed):
    g = random_graph(seed)
    desc = g.description()
    n, frames, blocks = 5, 200, 4
    freqs = np.array([82.41, 146.83, 220.0, 329.63, 523.25], dtype=np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    on = [3 + 11 * v for v in range(n)]
    off = [430 + 7 * v for v in range(n)]
    for v in range(n):
        eng.schedule_voice_event("gate", v, on[v], 0.9)
        eng.schedule_voice_event("gate", v, off[v], 0.0)
    voices = [VoiceInterp(desc, SR, {"frequency": float(freqs[v])}) for v in range(n)]
    got, ref = [], np.zeros((n, frames * blocks), dtype=np.float32)
    for b in range(blocks):
        if b == 1:
            eng.set_value("cutoff", 2600.0)
            for vi in voices:
                vi.set_value("cutoff", 2600.0)
        if b == 2:
            eng.set_value("amount", 0.2)
            for vi in voices:
                vi.set_value("amount", 0.2)
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
        for v, vi in enumerate(voices):
            for i in range(frames):
                f = b * frames + i
                gates = [("gate", 0.9)] if f == on[v] else ([("gate", 0.0)] if f == off[v] else [])
                ref[v, f] = vi.frame(gates)
    got = np.concatenate(got, axis=1)
    assert np.isfinite(ref).all()
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5, (seed, err, desc["order"])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_multirate_graph_matches_the_node_interpreter(seed):
    """Oversampled regions: `* N` nodes, Up/Down edges with every policy (and the default), value ports and
    explicit [latch] edges that stay un-resampled, outer nodes downstream of the region (row a4, a18-a20)."""
    from tests.test_codegen_fuzz_cpu import random_multirate_graph
    g = random_multirate_graph(seed)
    desc = g.description()
    n, frames, blocks = 4, 160, 3
    freqs = np.array([98.0, 220.0, 587.33, 1318.5], dtype=np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    on = [2 + 9 * v for v in range(n)]
    off = [300 + 5 * v for v in range(n)]
    for v in range(n):
        eng.schedule_voice_event("gate", v, on[v], 0.8)
        eng.schedule_voice_event("gate", v, off[v], 0.0)
    voices = [VoiceInterp(desc, SR, {"frequency": float(freqs[v])}) for v in range(n)]
    got, ref = [], np.zeros((n, frames * blocks), dtype=np.float32)
    for b in range(blocks):
        if b == 1:
            eng.set_value("cutoff", 4200.0)
            for vi in voices:
                vi.set_value("cutoff", 4200.0)
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
        for v, vi in enumerate(voices):
            for i in range(frames):
                f = b * frames + i
                gates = [("gate", 0.8)] if f == on[v] else ([("gate", 0.0)] if f == off[v] else [])
                ref[v, f] = vi.frame(gates)
    got = np.concatenate(got, axis=1)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-3
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5, (seed, err, desc["order"], desc["rates"], desc["policies"])


# ---- random CONNECTION EXPRESSIONS (round 3): frames, frame constructors, channel extraction, named functions, f32
# ---- methods, several outputs, outputs read as sources -- generated as trees, rendered to graph! text for the engine and
# ---- evaluated in numpy f32 over the oracle's oscillators for the reference -------------------------------------------
def _rand_scalar(rng, n_src, d=0):
    r = rng.random()
    if d > 3 or r < 0.25:
        return ("src", int(rng.integers(0, n_src)))
    if r < 0.33:
        return ("num", float(np.float32(rng.uniform(-1.5, 1.5))))
    if r < 0.6:
        return ("bin", "+-*"[int(rng.integers(0, 3))], _rand_scalar(rng, n_src, d + 1), _rand_scalar(rng, n_src, d + 1))
    if r < 0.8:
        m = ["tanh", "abs", "clamp", "max", "min", "sin", "powi", "half", "exp"][int(rng.integers(0, 9))]
        return ("fn", m, _rand_scalar(rng, n_src, d + 1))
    return ("chan", _rand_frame(rng, n_src, d + 1), int(rng.integers(0, 2)))


def _rand_frame(rng, n_src, d=0):
    r = rng.random()
    if d > 3 or r < 0.45:
        return ("frame", _rand_scalar(rng, n_src, d + 1), _rand_scalar(rng, n_src, d + 1))
    if r < 0.65:
        return ("swap", _rand_frame(rng, n_src, d + 1))
    if r < 0.8:
        return ("fscale", _rand_frame(rng, n_src, d + 1), float(np.float32(rng.uniform(-1.0, 1.0))))
    return ("fbin", "+-"[int(rng.integers(0, 2))], _rand_frame(rng, n_src, d + 1), _rand_frame(rng, n_src, d + 1))


def _text(e):
    k = e[0]
    if k == "src":
        return "o%d.output" % e[1]
    if k == "num":
        return repr(float(e[1])) if e[1] >= 0 else "(0.0 - %r)" % (-float(e[1]))
    if k == "bin":
        return "(%s %s %s)" % (_text(e[2]), e[1], _text(e[3]))
    if k == "fn":
        a = _text(e[2])
        return {"tanh": "%s.tanh()", "abs": "%s.abs()", "clamp": "%s.clamp(-0.6, 0.7)", "max": "%s.max(0.05)", "min": "%s.min(0.4)",
                "sin": "%s.sin()", "powi": "%s.powi(2)", "half": "fx::half(%s)", "exp": "%s.clamp(-4.0, 2.0).exp()"}[e[1]] % (("(%s)" % a) if e[1] != "half" else a)
    if k == "chan":
        return "%s[%d]" % (_text(e[1]), e[2])
    if k == "frame":
        return "Frame::<2>(%s, %s)" % (_text(e[1]), _text(e[2]))
    if k == "swap":
        return "swap2(%s)" % _text(e[1])
    if k == "fscale":
        return "(%s * %r)" % (_text(e[1]), float(e[2])) if e[2] >= 0 else "(%s * (0.0 - %r))" % (_text(e[1]), -float(e[2]))
    return "(%s %s %s)" % (_text(e[2]), e[1], _text(e[3]))


def _value(e, src):
    f = np.float32
    k = e[0]
    with np.errstate(all="ignore"):
        if k == "src":
            return f(src[e[1]])
        if k == "num":
            return f(e[1]) if e[1] >= 0 else f(f(0.0) - f(-e[1]))
        if k == "bin":
            a, b = _value(e[2], src), _value(e[3], src)
            return f({"+": a + b, "-": a - b, "*": a * b}[e[1]])
        if k == "fn":
            a = _value(e[2], src)
            return f({"tanh": lambda: np.tanh(a), "abs": lambda: np.abs(a), "clamp": lambda: np.clip(a, f(-0.6), f(0.7)),
                      "max": lambda: np.maximum(a, f(0.05)), "min": lambda: np.minimum(a, f(0.4)), "sin": lambda: np.sin(a),
                      "powi": lambda: a * a, "half": lambda: a * f(0.5), "exp": lambda: np.exp(np.clip(a, f(-4.0), f(2.0)))}[e[1]]())
        if k == "chan":
            return _value(e[1], src)[e[2]]
        if k == "frame":
            return (_value(e[1], src), _value(e[2], src))
        if k == "swap":
            a = _value(e[1], src)
            return (a[1], a[0])
        if k == "fscale":
            a = _value(e[1], src)
            c = f(e[2]) if e[2] >= 0 else f(f(0.0) - f(-e[2]))
            return (f(a[0] * c), f(a[1] * c))
        a, b = _value(e[2], src), _value(e[3], src)
        return (f(a[0] + b[0]), f(a[1] + b[1])) if e[1] == "+" else (f(a[0] - b[0]), f(a[1] - b[1]))


def random_expression_graph(seed):
    rng = np.random.default_rng(1000 + seed)
    n_src = int(rng.integers(2, 4))
    waves = [["saw", "sine", "square", "triangle"][int(rng.integers(0, 4))] for _ in range(n_src)]
    ratios = [float(np.float32(rng.uniform(0.5, 3.0))) for _ in range(n_src)]
    outs, budget = [], 4
    for _ in range(int(rng.integers(1, 4))):
        width = 2 if (rng.random() < 0.5 and budget >= 2) else 1
        if budget < width:
            break
        budget -= width
        outs.append((_rand_frame(rng, n_src) if width == 2 else _rand_scalar(rng, n_src), width))
    gain_in = _rand_scalar(rng, n_src)  # an expression feeding a NODE, whose output joins the last scalar output
    nodes = " ".join("o%d = PolyBlepOscillator::%s(220.0, 0.6);" % (i, w) for i, w in enumerate(waves)) + " g = Gain::new(0.75);"
    conns = " ".join("frequency * %r -> o%d.frequency;" % (r, i) for i, r in enumerate(ratios)) + " %s -> g.input;" % _text(gain_in)
    decl = ""
    # the node's output joins the last scalar output INSIDE its expression: a summed fan-in takes plain endpoints only (the
    # reference refuses a compound source there, codegen/emit_node.rs:90-93 -- and so does og_graph.cpp)
    last = max(i for i, (_, w) in enumerate(outs) if w == 1) if any(w == 1 for _, w in outs) else None
    for i, (e, w) in enumerate(outs):
        decl += "output out%d: stream%s; " % (i, ": Frame<2>" if w == 2 else "")
        conns += (" (%s) + g.output -> out%d;" if i == last else " %s -> out%d;") % (_text(e), i)
    text = "name: Fx%d; input frequency: value = 220.0; %s nodes { %s } connections { %s }" % (seed, decl, nodes, conns)
    return text, dict(waves=waves, ratios=ratios, outs=outs, gain_in=gain_in, last=last)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_random_connection_expressions_match_numpy_over_the_oracle_oscillators(seed):
    import ctypes as C

    from tests import oracle_lib as ol

    lib = ol.load()
    oscen_amd.register_function("fx::half", ["x"], "return x * 0.5f;")
    oscen_amd.register_function("swap2", [("v", 2)], "og::Frame<2> o; o.v[0] = v.v[1]; o.v[1] = v.v[0]; return o;", result_channels=2)
    try:
        text, info = random_expression_graph(seed)
        g = oscen_amd.Graph(dsl=text, per_voice=["frequency"])
        n, frames = 4, 300
        freqs = np.array([98.0, 220.0, 587.33, 1318.5], dtype=np.float32)
        eng = oscen_amd.Engine(g, n, sample_rate=SR)
        eng.set_voice_values("frequency", freqs)
        eng.set_voice_taps(list(range(n)))
        eng.process_block(frames)
        got = eng.read_voice_taps(frames)
        channels = sum(w for _, w in info["outs"])
        got = got.reshape(n, frames, channels)
        wave_id = {"sine": ol.PB_SINE, "saw": ol.PB_SAW, "square": ol.PB_SQUARE, "triangle": ol.PB_TRIANGLE}
        worst = 0.0
        for v in range(n):
            oscs = []
            for w, r in zip(info["waves"], info["ratios"]):
                o = ol.PolyBlep()
                lib.oo_polyblep_new(C.byref(o), float(np.float32(freqs[v]) * np.float32(r)), 0.6, wave_id[w])
                o.sample_rate = SR
                oscs.append(o)
            for i in range(frames):
                for o in oscs:
                    lib.oo_polyblep_process(C.byref(o))
                src = [np.float32(o.output) for o in oscs]
                gout = np.float32(_value(info["gain_in"], src) * np.float32(0.75))
                c = 0
                for oi, (e, w) in enumerate(info["outs"]):
                    val = _value(e, src)
                    vals = list(val) if w == 2 else [np.float32(val + gout) if oi == info["last"] else val]
                    for x in vals:
                        ref = float(x)
                        assert np.isfinite(ref), (seed, text)
                        worst = max(worst, abs(float(got[v, i, c]) - ref) / max(1.0, abs(ref)))
                        c += 1
        observed.note(worst)
        assert worst <= 1e-5, (seed, worst, text)
    finally:
        oscen_amd.unregister_function("fx::half")
        oscen_amd.unregister_function("swap2")
