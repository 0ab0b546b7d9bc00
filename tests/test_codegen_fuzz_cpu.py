"""Random voice graphs through the graph compiler and hiprtc (compile only, gfx950, no GPU needed): every
description the node registry allows must lower to a translation unit that compiles -- ordinary kernel,
2- and 4-wave pipelines, feedback edges, delay lines, oversampled regions."""
import numpy as np
import pytest

import oscen_amd

SOURCES = [("PolyBlepOscillator::saw", (220.0, 0.5)), ("PolyBlepOscillator::square", (220.0, 0.4)),
           ("PolyBlepOscillator::triangle", (220.0, 0.4)), ("PolyBlepOscillator::sine", (220.0, 0.4)),
           ("Oscillator::sine", (220.0, 0.5)), ("Oscillator::saw", (220.0, 0.5)), ("FmOperator::new", ())]
FILTERS = [("TptFilter::new", (1200.0, 0.8)), ("IirLowpass::new", (2000.0, 0.7)), ("LP18Filter::new", (900.0, 0.3)),
           ("Gain::new", (0.7,)), ("HardClip::new", ()), ("AddValue::new", (0.1,))]


class RecGraph(oscen_amd.Graph):
    """A Graph that also keeps its description as plain data (for tests/graph_interp.py)."""

    def __init__(self, name):
        super().__init__(name)
        self.desc = {"inputs": [], "nodes": [], "edges": [], "rates": {}, "policies": {}}

    def input_value(self, name, default=0.0, ramp=0, per_voice=False):
        self.desc["inputs"].append((name, "value", default, ramp))
        return super().input_value(name, default, ramp=ramp, per_voice=per_voice)

    def input_event(self, name):
        self.desc["inputs"].append((name, "event", 0.0, 0))
        return super().input_event(name)

    def node(self, name, type_ctor, *args, rate=1):
        self.desc["nodes"].append((name, type_ctor, args))
        if rate > 1:
            self.desc["rates"][name] = rate
        return super().node(name, type_ctor, *args, rate=rate)

    def connect(self, src, dst, policy=""):
        if policy:
            self.desc["policies"][len(self.desc["edges"])] = policy
        self.desc["edges"].append((src, dst))
        return super().connect(src, dst, policy)

    def connect_via(self, src, via, dst):
        self.desc["edges"] += [(src, "%s.input" % via), ("%s.output" % via, dst)]
        return super().connect_via(src, via, dst)

    def description(self):
        import re
        d = dict(self.desc)
        d["order"] = re.search(r"// Node order: (.*)", self.kernel_source()).group(1).split()
        return d


def random_graph(seed):
    rng = np.random.default_rng(seed)
    g = RecGraph("fuzz%d" % seed)
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_value("cutoff", 1500.0, ramp=int(rng.integers(0, 2)) * 480)
    g.input_value("amount", 0.5)
    g.input_event("gate")
    g.output_stream("out")
    n_src = int(rng.integers(1, 4))
    outs = []
    for i in range(n_src):
        t, args = SOURCES[int(rng.integers(0, len(SOURCES)))]
        g.node("s%d" % i, t, *args)
        if t.startswith("FmOperator"):
            g.connect("frequency", "s%d.base_freq" % i)
            if outs and rng.random() < 0.7:
                g.connect(outs[-1] + " * amount", "s%d.phase_mod" % i)
        else:
            g.connect("frequency" if rng.random() < 0.6 else "frequency * 2.0", "s%d.frequency" % i)
        outs.append("s%d.output" % i)
    n_env = int(rng.integers(0, 3))
    for i in range(n_env):
        g.node("e%d" % i, "AdsrEnvelope::new", 0.01, 0.1, 0.6, 0.2)
        g.connect("gate", "e%d.gate" % i)
    sig = " + ".join(outs)
    if n_env:
        sig = "(%s) * e0.output" % sig if len(outs) > 1 else "%s * e0.output" % sig
    for i in range(int(rng.integers(0, 4))):
        t, args = FILTERS[int(rng.integers(0, len(FILTERS)))]
        g.node("f%d" % i, t, *args)
        g.connect(sig, "f%d.input" % i)
        if t in ("TptFilter::new", "IirLowpass::new", "LP18Filter::new"):
            mod = "cutoff + e%d.output * 800.0" % (n_env - 1) if (n_env and rng.random() < 0.5 and t == "TptFilter::new") else "cutoff"
            g.connect(mod, "f%d.cutoff" % i)
        sig = "f%d.output" % i
    if rng.random() < 0.4:  # a feedback echo around the tail
        g.node("mix", "Mixer::new")
        g.node("fb", "Gain::new", 0.3)
        g.node("dl", "Delay::new", float(rng.integers(40, 4000)), 0.0)
        g.connect(sig, "mix.input_a").connect("fb.output", "mix.input_b")
        g.connect_via("mix.output", "dl", "fb.input")
        sig = "mix.output"
    g.connect(sig, "out")
    return g


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_graphs_lower_and_compile(seed):
    g = random_graph(seed)
    src = g.kernel_source()
    assert "voice_block" in src
    # the DSL printer round-trips the description to the same kernel
    g2 = oscen_amd.Graph(dsl=g.to_dsl(), per_voice=["frequency"])
    assert g2.kernel_source() == src
    assert g.jit_check() > 10000  # bytes of gfx950 code object


POLICIES = ["", "sinc", "sinc_iir", "linear", "latch"]


def random_multirate_graph(seed):
    """outer source(s) -> [up policy] -> oversampled chain -> [down policy] -> (outer filter) -> out"""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([2, 4, 8]))
    g = RecGraph("mr%d" % seed)
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_value("cutoff", 2500.0)
    g.input_event("gate")
    g.output_stream("out")
    waves = ["PolyBlepOscillator::saw", "PolyBlepOscillator::sine", "PolyBlepOscillator::square", "Oscillator::sine"]
    g.node("o0", waves[int(rng.integers(0, 4))], 220.0, 0.6)
    g.connect("frequency", "o0.frequency")
    src = "o0.output"
    if rng.random() < 0.5:
        g.node("o1", waves[int(rng.integers(0, 4))], 220.0, 0.3)
        g.connect("frequency * 2.0", "o1.frequency")
        src = "o0.output + o1.output"
    have_env = rng.random() < 0.6
    if have_env:
        g.node("env", "AdsrEnvelope::new", 0.003, 0.02, 0.6, 0.03)
        g.connect("gate", "env.gate")
    up = POLICIES[int(rng.integers(0, 5))]
    sig, first = src, True
    for i in range(int(rng.integers(1, 4))):
        kind = ["HardClip::new", "Gain::new", "TptFilter::new", "Vca::new"][int(rng.integers(0, 4))]
        if kind == "Vca::new" and not have_env:
            kind = "HardClip::new"
        args = {"HardClip::new": (), "Gain::new": (1.4,), "TptFilter::new": (3000.0, 0.8), "Vca::new": ()}[kind]
        g.node("i%d" % i, kind, *args, rate=N)
        g.connect(sig, "i%d.input" % i, up if first else "")
        if kind == "Vca::new":
            g.connect("env.output", "i%d.control" % i, POLICIES[int(rng.integers(0, 5))])
        if kind == "TptFilter::new":
            g.connect("cutoff", "i%d.cutoff" % i)
        sig, first = "i%d.output" % i, False
    down = POLICIES[int(rng.integers(0, 5))]
    if rng.random() < 0.5:
        kind = ["TptFilter::new", "IirLowpass::new", "Gain::new"][int(rng.integers(0, 3))]
        args = {"TptFilter::new": (1800.0, 0.9), "IirLowpass::new": (2200.0, 0.7), "Gain::new": (0.9,)}[kind]
        g.node("post", kind, *args)
        g.connect(sig, "post.input", down)
        g.connect("post.output" if not have_env or rng.random() < 0.5 else "post.output * env.output", "out")
    else:
        g.connect(sig, "out", down)
    return g


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_multirate_graphs_lower_and_compile(seed):
    g = random_multirate_graph(seed)
    src = g.kernel_source()
    assert "oversampled inner loop" in src
    assert oscen_amd.Graph(dsl=g.to_dsl(), per_voice=["frequency"]).kernel_source() == src
    assert g.jit_check() > 10000
