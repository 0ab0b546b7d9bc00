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
        self.v = {"input": f32(0), "input_a": f32(0), "input_b": f32(0), "gain": f32(args[0] if kind == "Gain" and args else 1.0),
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


class VoiceInterp:
    """One voice of a description (the dict built by tests: inputs, nodes, edges, order)."""

    def __init__(self, desc, sr, voice_values):
        self.lib = ol.load()
        self.desc = desc
        self.sr = sr
        self.nodes = {n: _make(self.lib, f32(sr), t, [float(a) for a in args]) for n, t, args in desc["nodes"]}
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
        self.edges = {}
        self.event_edges = {}
        kinds = {name: kind for name, kind, _, _ in desc["inputs"]}
        self.out_exprs = []
        for src, dst in desc["edges"]:
            if kinds.get(src.strip()) == "event":
                self.event_edges.setdefault(src.strip(), []).append(dst.split(".")[0])
            elif "." in dst:
                node, port = dst.split(".")
                self.edges.setdefault(node, {}).setdefault(port, []).append(_parse(src))
            else:
                self.out_exprs.append(_parse(src))
        self.last = {}

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

    def frame(self, gates=()):
        """one sample; gates: [(event input name, value)] delivered on this frame"""
        for r in self.ramps.values():  # tick_ramps (codegen/mod.rs:878-914)
            self.lib.oo_ramp_tick(C.byref(r.r))
        for name in self.desc["order"]:
            node = self.nodes[name]
            for port, srcs in self.edges.get(name, {}).items():
                acc = self._ev(srcs[0])
                for s in srcs[1:]:
                    acc = f32(acc + self._ev(s))
                node.set(port, acc)
            for ev_name, v in gates:
                if name in self.event_edges.get(ev_name, ()):
                    node.gate(v)
            node.process()
        out = self._ev(self.out_exprs[0])
        for e in self.out_exprs[1:]:
            out = f32(out + self._ev(e))
        return out
