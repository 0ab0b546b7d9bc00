"""IirLowpass, LP18Filter, Delay (+ feedback edges through a Delay).

The HIP path (hiprtc-compiled graphs, called through the C ABI) against the oracle nodes driven
sample by sample.  Tolerance: 1e-5 * max(1, |ref|) per sample.
"""
import ctypes as C

import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

SR = 48000.0
TOL = 1e-5


def _run(graph, setup, ref_fn, n, frames=256, blocks=3, per_block=None):
    eng = oscen_amd.Engine(graph, n, sample_rate=SR)
    setup(eng)
    eng.set_voice_taps(list(range(n)))
    got = []
    for b in range(blocks):
        if per_block:
            per_block(eng, b)
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
    got = np.concatenate(got, axis=1)
    ref = np.stack([ref_fn(v, frames, blocks) for v in range(n)])
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    return err, got, ref, eng


def _saw(lib, freq, amp=0.5):
    o = ol.PolyBlep()
    lib.oo_polyblep_new(C.byref(o), 440.0, amp, ol.PB_SAW)
    o.sample_rate = SR
    o.frequency = float(freq)
    return o


@pytest.mark.parametrize("per_voice_cutoff", [False, True])
def test_iir_lowpass(per_voice_cutoff):
    """saw -> IirLowpass.  Uniform cutoff: coefficients are host-derived slots, refreshed every 32nd tick
    (a cutoff change between blocks is picked up exactly where the reference picks it up).  Per-voice
    cutoff (`frequency * 3.0 -> filt.cutoff`): the coefficient formula runs on the device."""
    lib = ol.load()
    n = 20
    freqs = np.geomspace(55.0, 3520.0, n).astype(np.float32)
    g = oscen_amd.Graph("iir_lp_%d" % per_voice_cutoff)
    g.input_value("frequency", 440.0, per_voice=True)
    g.input_value("cutoff", 900.0)
    g.input_value("q", 2.5)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("filt", "IirLowpass::new", 1000.0, 0.707)
    g.connect("frequency", "osc.frequency").connect("osc.output", "filt.input")
    g.connect("frequency * 3.0" if per_voice_cutoff else "cutoff", "filt.cutoff").connect("q", "filt.q")
    g.connect("filt.output", "out")
    cut = [900.0, 900.0, 4000.0]

    def ref(v, frames, blocks):
        o = _saw(lib, freqs[v])
        f = ol.IirLowpass()
        lib.oo_iir_lowpass_new(C.byref(f), 1000.0, 0.707)
        f.sample_rate = SR
        lib.oo_iir_lowpass_prepare(C.byref(f))
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            lib.oo_polyblep_process(C.byref(o))
            f.cutoff = float(np.float32(freqs[v]) * np.float32(3.0)) if per_voice_cutoff else cut[i // frames]
            f.q = 2.5
            f.input = o.output
            lib.oo_iir_lowpass_process(C.byref(f))
            out[i] = f.output
        return out

    err, got, r, _ = _run(g, lambda e: e.set_voice_values("frequency", freqs), ref, n, frames=250,  # 250: the 32-tick
                          per_block=lambda e, b: e.set_value("cutoff", cut[b]))                    # counter straddles blocks
    assert np.max(np.abs(r)) > 0.2
    observed.note(err)
    assert err <= TOL, err


def test_lp18_filter_with_envelope_fmod():
    """saw -> LP18Filter, cutoff uniform, fmod from the envelope (value input fed by a stream expression),
    resonance changed between blocks (the change-detection fields are per-voice state)."""
    lib = ol.load()
    n = 12
    freqs = np.geomspace(55.0, 1760.0, n).astype(np.float32)
    g = oscen_amd.Graph("lp18_voice")
    g.input_value("frequency", 440.0, per_voice=True)
    g.input_value("cutoff", 700.0)
    g.input_value("res", 0.3)
    g.input_event("gate")
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("env", "AdsrEnvelope::new", 0.005, 0.05, 0.6, 0.1)
    g.node("filt", "LP18Filter::new", 500.0, 1.5)  # ctor resonance above the clamp on purpose
    g.connect("frequency", "osc.frequency").connect("gate", "env.gate").connect("osc.output", "filt.input")
    g.connect("cutoff", "filt.cutoff").connect("env.output * 3000.0", "filt.fmod").connect("res", "filt.resonance")
    g.connect("filt.output", "out")
    res = [0.3, 0.85, 2.0]

    def ref(v, frames, blocks):
        o = _saw(lib, freqs[v])
        e = ol.Adsr()
        lib.oo_adsr_new(C.byref(e), 0.005, 0.05, 0.6, 0.1)
        e.sample_rate = SR
        lib.oo_adsr_prepare(C.byref(e))
        f = ol.Lp18()
        lib.oo_lp18_new(C.byref(f), 500.0, 1.5)
        f.sample_rate = SR
        lib.oo_lp18_prepare(C.byref(f))
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            if i == 7 + v:
                ev = ol.Event(0, 0.9, 0)
                lib.oo_adsr_handle_gate_event(C.byref(e), C.byref(ev))
            lib.oo_polyblep_process(C.byref(o))
            lib.oo_adsr_process(C.byref(e))
            f.cutoff = 700.0
            f.fmod = float(np.float32(e.output) * np.float32(3000.0))
            f.resonance = res[i // frames]
            f.input = o.output
            lib.oo_lp18_process(C.byref(f))
            out[i] = f.output
        return out

    def setup(eng):
        eng.set_voice_values("frequency", freqs)
        for v in range(n):
            eng.schedule_voice_event("gate", v, 7 + v, 0.9)

    err, got, r, _ = _run(g, setup, ref, n, per_block=lambda e, b: e.set_value("res", res[b]))
    assert np.max(np.abs(r)) > 0.05
    observed.note(err)
    assert err <= TOL, err


def test_lp18_unconnected_parameters_match_prepare():
    lib = ol.load()
    n = 4
    freqs = np.array([110, 220, 440, 880], dtype=np.float32)
    g = oscen_amd.Graph("lp18_plain")
    g.input_value("frequency", 440.0, per_voice=True)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("filt", "LP18Filter::new", 1200.0, 1.5)
    g.connect("frequency", "osc.frequency").connect("osc.output", "filt.input").connect("filt.output", "out")

    def ref(v, frames, blocks):
        o = _saw(lib, freqs[v])
        f = ol.Lp18()
        lib.oo_lp18_new(C.byref(f), 1200.0, 1.5)
        f.sample_rate = SR
        lib.oo_lp18_prepare(C.byref(f))
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            lib.oo_polyblep_process(C.byref(o))
            f.input = o.output
            lib.oo_lp18_process(C.byref(f))
            out[i] = f.output
        return out

    err, _, r, _ = _run(g, lambda e: e.set_voice_values("frequency", freqs), ref, n, blocks=2)
    observed.note(err)
    assert err <= TOL, err


def _delay(lib, samples, fb):
    d = ol.Delay()
    lib.oo_delay_new(C.byref(d), samples, fb)
    d.sample_rate = SR
    lib.oo_delay_prepare(C.byref(d))
    return d


@pytest.mark.parametrize("delay_samples,feedback", [(100.0, 0.0), (37.5, 0.6), (300.25, 1.7), (0.0, 0.5)])
def test_delay_line(delay_samples, feedback):
    """saw -> Delay(delay_samples, feedback): whole-sample reads (exact sample), fractional reads
    (Catmull-Rom over four ring slots), the 0.99 feedback clamp, and delays both shorter and longer
    than a block, so reads hit samples written by the same launch and by earlier launches."""
    lib = ol.load()
    n = 70  # more than one wave, ragged
    freqs = np.geomspace(55.0, 1760.0, n).astype(np.float32)
    g = oscen_amd.Graph("delay_%d" % int(delay_samples * 4))
    g.input_value("frequency", 440.0, per_voice=True)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("d", "Delay::new", delay_samples, feedback)
    g.connect("frequency", "osc.frequency").connect("osc.output", "d.input").connect("d.output + osc.output", "out")

    def ref(v, frames, blocks):
        o = _saw(lib, freqs[v])
        d = _delay(lib, delay_samples, feedback)
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            lib.oo_polyblep_process(C.byref(o))
            d.input = o.output
            lib.oo_delay_process(C.byref(d))
            out[i] = np.float32(d.output) + np.float32(o.output)
        lib.oo_delay_free(C.byref(d))
        return out

    err, got, r, eng = _run(g, lambda e: e.set_voice_values("frequency", freqs), ref, n, frames=128, blocks=5)
    observed.note(err)
    assert err <= TOL, err
    assert eng.state_bytes >= 131072 * n * 4  # the delay lines are part of the saved state


def test_modulated_delay_time():
    """Chorus-style: an LFO sweeps delay_samples every frame (fractional reads at moving positions);
    the every-32nd-tick clamp only touches the frame it runs on (delay/mod.rs:47-56)."""
    lib = ol.load()
    n = 16
    freqs = np.geomspace(110.0, 1760.0, n).astype(np.float32)
    g = oscen_amd.Graph("chorus")
    g.input_value("frequency", 440.0, per_voice=True)
    g.input_value("depth", 40.0)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("lfo", "Oscillator::sine", 3.0, 1.0)
    g.node("d", "Delay::new", 50.0, 0.2)
    g.connect("frequency", "osc.frequency").connect("osc.output", "d.input")
    g.connect("lfo.output * depth + 30.0", "d.delay_samples").connect("d.output", "out")

    def ref(v, frames, blocks):
        o = _saw(lib, freqs[v])
        l = ol.Oscillator()
        lib.oo_oscillator_new(C.byref(l), 3.0, 1.0, ol.WAVE_SINE)
        l.sample_rate = SR
        d = _delay(lib, 50.0, 0.2)
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            lib.oo_polyblep_process(C.byref(o))
            lib.oo_oscillator_process(C.byref(l))
            d.delay_samples = float(np.float32(l.output) * np.float32(40.0) + np.float32(30.0))  # goes negative: clamped by get()
            d.input = o.output
            lib.oo_delay_process(C.byref(d))
            out[i] = d.output
        lib.oo_delay_free(C.byref(d))
        return out

    err, _, r, _ = _run(g, lambda e: e.set_voice_values("frequency", freqs), ref, n, frames=256, blocks=3)
    assert np.max(np.abs(r)) > 0.2
    observed.note(err)
    assert err <= TOL, err


def test_feedback_edge_through_delay_node():
    """`mix.output -> [d] -> fbk.input; fbk.output -> mix.input_b`: fbk is scheduled before d and reads
    the sample d produced on the previous frame; an inline `-> [3] ->` delay feeds the output."""
    lib = ol.load()
    text = """
    name: EchoVoice;
    input frequency: value = 220.0;
    input gate: event;
    input fb: value = 0.5;
    output out: stream;
    nodes {
        osc = PolyBlepOscillator::saw(220.0, 0.5);
        env = AdsrEnvelope::new(0.002, 0.03, 0.0, 0.05);
        mix = Mixer::new();
        fbk = Gain::new(0.4);
        d = Delay::new(90.0, 0.0);
    }
    connections {
        frequency -> osc.frequency;
        gate -> env.gate;
        fb -> fbk.gain;
        osc.output * env.output -> mix.input_a;
        mix.output -> [d] -> fbk.input;
        fbk.output -> mix.input_b;
        mix.output -> [3] -> out;
    }
    """
    g = oscen_amd.Graph(dsl=text, per_voice=["frequency"])
    n = 9
    freqs = np.geomspace(110.0, 1760.0, n).astype(np.float32)

    def ref(v, frames, blocks):
        o = _saw(lib, freqs[v])
        e = ol.Adsr()
        lib.oo_adsr_new(C.byref(e), 0.002, 0.03, 0.0, 0.05)
        e.sample_rate = SR
        lib.oo_adsr_prepare(C.byref(e))
        d = _delay(lib, 90.0, 0.0)
        d3 = _delay(lib, 3.0, 0.0)
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            if i == 5:
                ev = ol.Event(0, 1.0, 0)
                lib.oo_adsr_handle_gate_event(C.byref(e), C.byref(ev))
            lib.oo_polyblep_process(C.byref(o))
            lib.oo_adsr_process(C.byref(e))
            fbk = np.float32(d.output) * np.float32(0.5)  # d.output: still last frame's value
            mix = np.float32(o.output) * np.float32(e.output) + fbk
            d.input = float(mix)
            lib.oo_delay_process(C.byref(d))
            d3.input = float(mix)
            lib.oo_delay_process(C.byref(d3))
            out[i] = d3.output
        lib.oo_delay_free(C.byref(d))
        lib.oo_delay_free(C.byref(d3))
        return out

    def setup(eng):
        eng.set_voice_values("frequency", freqs)
        for v in range(n):
            eng.schedule_voice_event("gate", v, 5, 1.0)

    err, got, r, _ = _run(g, setup, ref, n, frames=256, blocks=4)
    # the echoes are there: energy after the envelope has died (0.03 s decay to sustain 0)
    assert np.max(np.abs(r[:, 600:])) > 1e-3
    observed.note(err)
    assert err <= TOL, err


def test_delay_state_roundtrip():
    """save_state / load_state carry the delay lines: a restored engine continues bit-identically."""
    g = oscen_amd.Graph("delay_state")
    g.input_value("frequency", 440.0, per_voice=True)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("d", "Delay::new", 200.5, 0.7)
    g.connect("frequency", "osc.frequency").connect("osc.output", "d.input").connect("d.output", "out")
    n = 8
    freqs = np.geomspace(110.0, 880.0, n).astype(np.float32)
    a = oscen_amd.Engine(g, n, sample_rate=SR)
    a.set_voice_values("frequency", freqs)
    for _ in range(3):
        a.process_block(256)
    blob = a.save_state()
    want = [a.process_block(256).copy() for _ in range(2)]
    b = oscen_amd.Engine(g, n, sample_rate=SR)
    b.load_state(blob)
    got = [b.process_block(256).copy() for _ in range(2)]
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
