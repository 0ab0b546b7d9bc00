"""A per-sample INTERPRETER of voice-graph descriptions over the oracle's nodes (test infrastructure).

It restates what the reference's generated `process()` does for one voice
(oscen-graph-compiler/src/codegen/emit_frame.rs:29-69, emit_node.rs:35-286): tick the ramped inputs,
then for every node in schedule order resolve its inputs (one source = copy, several = sum in edge
order, compound sources evaluated as f32 expressions; a source that has not run yet in this frame --
a feedback edge -- yields what it produced on the previous frame), deliver the frame's events, and
call the node's `process()`.  The nodes themselves are the C oracle's (ctypes).

Used to check the graph COMPILER (ordering, fan-in, expressions, feedback edges, hoisting, pipelines)
on descriptions nobody wrote a hand-expanded oracle for.
"""
import ctypes as C
import re

import numpy as np

from tests import oracle_lib as ol

f32 = np.float32


class _Node:
    outputs = ("output",)

    def __init__(self, lib, sr, args):
        self.lib = lib

    def set(self, port, v):
        setattr(self.s, port, float(v))

    def get(self, port):
        return f32(getattr(self.s, port))

    def gate(self, v):
        raise AssertionError("node has no event input")


class _PolyBlep(_Node):
    WAVES = {"sine": ol.PB_SINE, "saw": ol.PB_SAW, "square": ol.PB_SQUARE, "triangle": ol.PB_TRIANGLE}

    def __init__(self, lib, sr, args, wave):
        self.lib, self.s = lib, ol.PolyBlep()
        lib.oo_polyblep_new(C.byref(self.s), args[0], args[1], self.WAVES[wave])
        self.s.sample_rate = sr

    def process(self):
        self.lib.oo_polyblep_process(C.byref(self.s))


class _Oscillator(_Node):
    WAVES = {"sine": ol.WAVE_SINE, "square": ol.WAVE_SQUARE, "saw": ol.WAVE_SAW}

    def __init__(self, lib, sr, args, wave):
        self.lib, self.s = lib, ol.Oscillator()
        lib.oo_oscillator_new(C.byref(self.s), args[0], args[1], self.WAVES[wave])
        self.s.sample_rate = sr

    def process(self):
        self.lib.oo_oscillator_process(C.byref(self.s))


class _FmOperator(_Node):
    def __init__(self, lib, sr, args):
        self.lib, self.s = lib, ol.FmOperator()
        lib.oo_fm_operator_new(C.byref(self.s))
        self.s.sample_rate = sr

    def process(self):
        self.lib.oo_fm_operator_process(C.byref(self.s))


class _Adsr(_Node):
    def __init__(self, lib, sr, args):
        self.lib, self.s = lib, ol.Adsr()
        lib.oo_adsr_new(C.byref(self.s), *args)
        self.s.sample_rate = sr
        lib.oo_adsr_prepare(C.byref(self.s))

    def gate(self, v):
        ev = ol.Event(0, float(v), 0)
        self.lib.oo_adsr_handle_gate_event(C.byref(self.s), C.byref(ev))

    def process(self):
        self.lib.oo_adsr_process(C.byref(self.s))


class _Tpt(_Node):
    def __init__(self, lib, sr, args):
        self.lib, self.s = lib, ol.Tpt()
        lib.oo_tpt_new(C.byref(self.s), args[0], args[1], 1)
        self.s.sample_rate = sr
        lib.oo_tpt_prepare(C.byref(self.s))

    def set(self, port, v):
        if port == "input":
            self.s.input[0] = float(v)
        else:
            setattr(self.s, port, float(v))

    def get(self, port):
        return f32(self.s.output[0])

    def process(self):
        self.lib.oo_tpt_process(C.byref(self.s))


class _Iir(_Node):
    def __init__(self, lib, sr, args):
        self.lib, self.s = lib, ol.IirLowpass()
        lib.oo_iir_lowpass_new(C.byref(self.s), args[0], args[1])
        self.s.sample_rate = sr
        lib.oo_iir_lowpass_prepare(C.byref(self.s))

    def process(self):
        self.lib.oo_iir_lowpass_process(C.byref(self.s))


class _Lp18(_Node):
    def __init__(self, lib, sr, args):
        self.lib, self.s = lib, ol.Lp18()
        lib.oo_lp18_new(C.byref(self.s), args[0], args[1])
        self.s.sample_rate = sr
        lib.oo_lp18_prepare(C.byref(self.s))

    def process(self):
        self.lib.oo_lp18_process(C.byref(self.s))


class _Delay(_Node):
    def __init__(self, lib, sr, args):
        self.lib, self.s = lib, ol.Delay()
        lib.oo_delay_new(C.byref(self.s), args[0], args[1])
        self.s.sample_rate = sr
        lib.oo_delay_prepare(C.byref(self.s))

    def process(self):
        self.lib.oo_delay_process(C.byref(self.s))

    def __del__(self):
        self.lib.oo_delay_free(C.byref(self.s))


class _Py(_Node):
    """Stateless f32 nodes (gain/mod.rs, fm-synth nodes/*.rs, oversampled-saturator HardClip)."""

    def __init__(self, lib, sr, args, kind):
        self.kind = kind
        self.v = {"input": f32(0), "input_a": f32(0), "input_b": f32(0), "control": f32(1.0),
                  "gain": f32(args[0] if kind == "Gain" and args else 1.0),
                  "value": f32(args[0] if kind == "AddValue" and args else 0.0), "output": f32(0)}

    def set(self, port, v):
        self.v[port] = f32(v)

    def get(self, port):
        return self.v["output"]

    def process(self):
        v = self.v
        if self.kind == "Gain":
            v["output"] = v["input"] * v["gain"]
        elif self.kind == "AddValue":
            v["output"] = v["input"] + v["value"]
        elif self.kind == "Mixer":
            v["output"] = v["input_a"] + v["input_b"]
        elif self.kind == "Vca":
            v["output"] = v["input"] * v["control"]
        else:  # HardClip: (x * 1.5).clamp(-0.7, 0.7)
            v["output"] = f32(min(max(v["input"] * f32(1.5), f32(-0.7)), f32(0.7)))


def _make(lib, sr, type_ctor, args):
    t, ctor = type_ctor.split("::")
    if t == "PolyBlepOscillator":
        return _PolyBlep(lib, sr, args, ctor)
    if t == "Oscillator":
        return _Oscillator(lib, sr, args, ctor)
    table = {"FmOperator": _FmOperator, "AdsrEnvelope": _Adsr, "TptFilter": _Tpt, "IirLowpass": _Iir, "LP18Filter": _Lp18,
             "Delay": _Delay}
    if t in table:
        return table[t](lib, sr, args)
    return _Py(lib, sr, args, t)


_TOK = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][-+]?\d+)?)|([A-Za-z_]\w*(?:\.\w+)?)|(.))")


def _parse(expr):
    """source expression -> nested tuples ('num', v) | ('ref', name) | (op, a, b) | ('neg', a); usual precedence."""
    toks = [(m.group(1), m.group(2), m.group(3)) for m in _TOK.finditer(expr) if m.group(0).strip()]
    pos = [0]

    def peek():
        return toks[pos[0]] if pos[0] < len(toks) else (None, None, None)

    def atom():
        num, ref, ch = peek()
        pos[0] += 1
        if num is not None:
            return ("num", f32(num))
        if ref is not None:
            return ("ref", ref)
        if ch == "(":
            e = add()
            pos[0] += 1  # ')'
            return e
        if ch == "-":
            return ("neg", atom())
        raise ValueError("bad expression: " + expr)

    def mul():
        a = atom()
        while peek()[2] in ("*", "/"):
            op = peek()[2]
            pos[0] += 1
            a = (op, a, atom())
        return a

    def add():
        a = mul()
        while peek()[2] in ("+", "-"):
            op = peek()[2]
            pos[0] += 1
            a = (op, a, mul())
        return a

    return add()


class _Up:
    """outer -> oversampled stream edge (emit_frame.rs:254-307): one kernel instance per edge"""

    def __init__(self, lib, policy, n):
        self.lib, self.policy, self.n = lib, policy or "sinc", n
        self.buf = np.zeros(n, dtype=f32)
        if self.policy == "sinc":
            self.st = ol.SincUp()
            lib.oo_sinc_up_new(C.byref(self.st), n)
        elif self.policy == "sinc_iir":
            self.st = ol.IirResampler()
            lib.oo_iir_resampler_new(C.byref(self.st), n)
        elif self.policy == "linear":
            self.st = ol.LinearUp()
            lib.oo_linear_up_new(C.byref(self.st), n)

    def push(self, x):
        x = float(x)
        if self.policy == "sinc":
            self.lib.oo_sinc_up_process(C.byref(self.st), x, ol.fptr(self.buf))
        elif self.policy == "sinc_iir":
            self.lib.oo_iir_up_process(C.byref(self.st), x, ol.fptr(self.buf))
        elif self.policy == "linear":
            self.lib.oo_linear_up_process(C.byref(self.st), x, ol.fptr(self.buf))
        else:
            self.lib.oo_latch_up_process(self.n, x, ol.fptr(self.buf))


class _Down:
    """oversampled -> outer stream edge (emit_frame.rs:474-514)"""

    def __init__(self, lib, policy, n):
        self.lib, self.policy, self.n = lib, policy or "sinc", n
        self.buf = np.zeros(n, dtype=f32)
        if self.policy == "sinc":
            self.st = ol.SincDown()
            lib.oo_sinc_down_new(C.byref(self.st), n)
        elif self.policy == "sinc_iir":
            self.st = ol.IirResampler()
            lib.oo_iir_resampler_new(C.byref(self.st), n)

    def pull(self):
        if self.policy == "sinc":
            return f32(self.lib.oo_sinc_down_process(C.byref(self.st), ol.fptr(self.buf)))
        if self.policy == "sinc_iir":
            return f32(self.lib.oo_iir_down_process(C.byref(self.st), ol.fptr(self.buf)))
        if self.policy == "linear":
            return f32(self.lib.oo_linear_down_process(self.n, ol.fptr(self.buf)))
        return f32(self.lib.oo_latch_down_process(self.n, ol.fptr(self.buf)))


# stream ('S') / value ('V') kind of the node inputs that matter for cross-rate edges (node definitions in the
# reference: #[input(stream)] / #[input(value)])
PORT_KIND = {("PolyBlepOscillator", "frequency"): "V", ("PolyBlepOscillator", "amplitude"): "V",
             ("PolyBlepOscillator", "pulse_width"): "V", ("Oscillator", "frequency"): "V", ("Oscillator", "amplitude"): "V",
             ("TptFilter", "q"): "V", ("IirLowpass", "cutoff"): "V", ("IirLowpass", "q"): "V", ("AddValue", "value"): "V",
             ("LP18Filter", "cutoff"): "V", ("LP18Filter", "fmod"): "V", ("LP18Filter", "resonance"): "V"}


def _refs(e, out):
    if e[0] == "ref":
        out.append(e[1])
    elif e[0] != "num":
        for sub in e[1:]:
            _refs(sub, out)
    return out


class VoiceInterp:
    """One voice of a description (the dict built by tests: inputs, nodes, edges, order; optional
    "rates": {node: N} for `* N` nodes and "policies": {edge index: policy} for cross-rate edges)."""

    def __init__(self, desc, sr, voice_values):
        self.lib = ol.load()
        self.desc = desc
        self.sr = sr
        rates = desc.get("rates", {})
        self.N = max([1] + list(rates.values()))
        self.inner = {n for n, r in rates.items() if r > 1}
        self.types = {n: t.split("::")[0] for n, t, _ in desc["nodes"]}
        self.nodes = {n: _make(self.lib, f32(sr * (self.N if n in self.inner else 1)), t, [float(a) for a in args])
                      for n, t, args in desc["nodes"]}
        self.values = {}
        self.ramps = {}
        self.active = C.c_uint32(0)
        for name, kind, default, ramp in desc["inputs"]:
            if kind != "value":
                continue
            self.values[name] = f32(voice_values.get(name, default))
            if ramp:
                r = ol.RampedInput()
                self.lib.oo_ramp_new(C.byref(r.r), float(default))
                r.default_frames = ramp
                self.ramps[name] = r
        kinds = {name: kind for name, kind, _, _ in desc["inputs"]}
        policies = desc.get("policies", {})
        # edge records: dict(expr, kind in plain/up/down, obj)
        self.edges = {}
        self.event_edges = {}
        self.out_edges = []
        node_deps = {n: set() for n in self.nodes}
        parsed = []
        for idx, (src, dst) in enumerate(desc["edges"]):
            if kinds.get(src.strip()) == "event":
                self.event_edges.setdefault(src.strip(), []).append(dst.split(".")[0])
                continue
            e = _parse(src)
            src_nodes = {r.split(".")[0] for r in _refs(e, []) if "." in r}
            parsed.append((idx, e, src_nodes, dst))
            if "." in dst:
                node_deps[dst.split(".")[0]] |= src_nodes
        # outer nodes downstream of the oversampled region run after the inner loop (emit_frame.rs:183-215)
        self.post = set()
        changed = True
        while changed:
            changed = False
            for n in desc["order"]:
                if n in self.inner or n in self.post:
                    continue
                if any(d in self.inner or d in self.post for d in node_deps[n]):
                    self.post.add(n)
                    changed = True
        self.ups, self.downs = [], []
        for idx, e, src_nodes, dst in parsed:
            src_inner = any(n in self.inner for n in src_nodes)
            src_outer = any(n not in self.inner for n in src_nodes)
            assert not (src_inner and src_outer), "expression mixes rates"
            pol = policies.get(idx, "")
            rec = {"expr": e, "kind": "plain", "obj": None}
            if "." in dst:
                node, port = dst.split(".")
                dst_inner = node in self.inner
                if dst_inner and src_outer and not src_inner:
                    if PORT_KIND.get((self.types[node], port), "S") == "S" and pol != "latch":
                        rec = {"expr": e, "kind": "up", "obj": _Up(self.lib, pol, self.N)}
                        self.ups.append(rec)
                elif (not dst_inner) and src_inner:
                    rec = {"expr": e, "kind": "down", "obj": _Down(self.lib, pol, self.N), "value": f32(0)}
                    self.downs.append(rec)
                self.edges.setdefault(node, {}).setdefault(port, []).append(rec)
            else:
                if src_inner:
                    rec = {"expr": e, "kind": "down", "obj": _Down(self.lib, pol, self.N), "value": f32(0)}
                    self.downs.append(rec)
                self.out_edges.append(rec)
        self.j = 0

    def set_value(self, name, v):
        if name in self.ramps:
            self.lib.oo_ramped_set(C.byref(self.ramps[name]), C.byref(self.active), float(v))
        else:
            self.values[name] = f32(v)

    def _ev(self, e):
        k = e[0]
        if k == "num":
            return e[1]
        if k == "ref":
            r = e[1]
            if "." in r:
                node, port = r.split(".")
                return self.nodes[node].get(port)
            return f32(self.ramps[r].r.current) if r in self.ramps else self.values[r]
        if k == "neg":
            return f32(-self._ev(e[1]))
        a, b = self._ev(e[1]), self._ev(e[2])
        with np.errstate(all="ignore"):
            return f32({"+": a + b, "-": a - b, "*": a * b, "/": a / b}[k])

    def _edge(self, rec):
        if rec["kind"] == "up":
            return f32(rec["obj"].buf[self.j])
        if rec["kind"] == "down":
            return rec["value"]
        return self._ev(rec["expr"])

    def _run(self, name, gates):
        node = self.nodes[name]
        for port, recs in self.edges.get(name, {}).items():
            acc = self._edge(recs[0])
            for r in recs[1:]:
                acc = f32(acc + self._edge(r))
            node.set(port, acc)
        for ev_name, v in gates:
            if name in self.event_edges.get(ev_name, ()):
                node.gate(v)
        node.process()

    def frame(self, gates=()):
        """one (outer-rate) sample; gates: [(event input name, value)] delivered on this frame"""
        for r in self.ramps.values():  # tick_ramps (codegen/mod.rs:878-914)
            self.lib.oo_ramp_tick(C.byref(r.r))
        order = self.desc["order"]
        for name in order:  # outer nodes that do not depend on the oversampled region
            if name not in self.inner and name not in self.post:
                self._run(name, gates)
        if self.inner:
            for rec in self.ups:  # upsample once per outer frame into a [N] buffer
                rec["obj"].push(self._ev(rec["expr"]))
            for j in range(self.N):
                self.j = j
                for name in order:
                    if name in self.inner:
                        self._run(name, gates if j == 0 else ())
                for rec in self.downs:  # capture every inner tick
                    rec["obj"].buf[j] = self._ev(rec["expr"])
            for rec in self.downs:  # downsample once per outer frame
                rec["value"] = rec["obj"].pull()
            for name in order:
                if name in self.post:
                    self._run(name, gates)
        out = self._edge(self.out_edges[0])
        for rec in self.out_edges[1:]:
            out = f32(out + self._edge(rec))
        return out
