"""A graph that was NOT compiled ahead of time goes through hiprtc at og_create()."""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_custom_graph_is_jit_compiled_and_matches_oracle_nodes():
    # README `Synth` shape (README.md:23-53): sine LFO -> saw frequency_mod -> TPT, per-voice carrier frequency
    g = oscen_amd.Graph("readme_synth")
    g.input_value("carrier_freq", 440.0, per_voice=True)
    g.input_value("mod_freq", 5.0)
    g.input_value("mod_depth", 0.2)
    g.input_value("cutoff", 1200.0)
    g.output_stream("audio_out")
    g.node("modulator", "PolyBlepOscillator::sine", 5.0, 0.2)
    g.node("carrier", "PolyBlepOscillator::saw", 440.0, 0.5)
    g.node("filter", "TptFilter::new", 1200.0, 0.707)
    g.connect("mod_freq", "modulator.frequency").connect("mod_depth", "modulator.amplitude")
    g.connect("carrier_freq", "carrier.frequency").connect("cutoff", "filter.cutoff")
    g.connect("modulator.output", "carrier.frequency_mod").connect("carrier.output", "filter.input")
    g.connect("filter.output", "audio_out")
    n, frames, sr = 80, 256, 48000.0
    eng = oscen_amd.Engine(g, n, sample_rate=sr)
    freqs = np.linspace(110.0, 1760.0, n).astype(np.float32)
    eng.set_voice_values("carrier_freq", freqs)
    eng.set_voice_taps(list(range(n)))
    import ctypes as C
    lib = ol.load()
    mods, cars, fils = [], [], []
    for v in range(n):
        m, c, f = ol.PolyBlep(), ol.PolyBlep(), ol.Tpt()
        lib.oo_polyblep_new(C.byref(m), 5.0, 0.2, ol.PB_SINE)
        lib.oo_polyblep_new(C.byref(c), 440.0, 0.5, ol.PB_SAW)
        lib.oo_tpt_new(C.byref(f), 1200.0, 0.707, 1)
        m.sample_rate = c.sample_rate = f.sample_rate = sr
        lib.oo_tpt_prepare(C.byref(f))
        mods.append(m); cars.append(c); fils.append(f)
    worst = 0.0
    for b in range(4):
        if b == 2:
            eng.set_value("cutoff", 2500.0)
        eng.process_block(frames)
        got = eng.read_voice_taps(frames)
        ref = np.zeros((n, frames), dtype=np.float32)
        for v in range(n):
            m, c, f = mods[v], cars[v], fils[v]
            for i in range(frames):
                lib.oo_polyblep_process(C.byref(m))
                c.frequency = float(freqs[v])
                c.frequency_mod = m.output
                lib.oo_polyblep_process(C.byref(c))
                f.cutoff = 2500.0 if b >= 2 else 1200.0
                f.input[0] = c.output
                lib.oo_tpt_process(C.byref(f))
                ref[v, i] = f.output[0]
        worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))))
    observed.note(worst)
    assert worst <= 1e-5, worst


def _run_nodes(engine_graph, per_voice_setup, ref_fn, n=24, frames=256, blocks=3, sr=48000.0):
    eng = oscen_amd.Engine(engine_graph, n, sample_rate=sr)
    per_voice_setup(eng)
    eng.set_voice_taps(list(range(n)))
    got = []
    for _ in range(blocks):
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
    got = np.concatenate(got, axis=1)
    ref = np.stack([ref_fn(v, frames * blocks) for v in range(n)])
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))), ref


@pytest.mark.parametrize("wave,ctor", [(ol.PB_SQUARE, "square"), (ol.PB_TRIANGLE, "triangle"),
                                       (ol.PB_SAW, "saw"), (ol.PB_SINE, "sine")])
def test_polyblep_waveforms(wave, ctor):
    # PolyBlepOscillator (oscillators/mod.rs:88-233): all four waveforms, incl. the >= sr/4 sine fallback
    import ctypes as C
    lib = ol.load()
    n = 24
    freqs = np.concatenate([np.geomspace(27.5, 11000.0, n - 2), [12000.0, 15000.0]]).astype(np.float32)
    g = oscen_amd.Graph("pb_" + ctor)
    g.input_value("frequency", 440.0, per_voice=True)
    g.input_value("pw", 0.5)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::" + ctor, 440.0, 0.8)
    g.connect("frequency", "osc.frequency").connect("pw", "osc.pulse_width").connect("osc.output", "out")

    def setup(eng):
        eng.set_voice_values("frequency", freqs)
        eng.set_value("pw", 0.3)

    def ref(v, total):
        o = ol.PolyBlep()
        lib.oo_polyblep_new(C.byref(o), 440.0, 0.8, wave)
        o.sample_rate = 48000.0
        o.frequency = float(freqs[v])
        o.pulse_width = 0.3
        out = np.zeros(total, dtype=np.float32)
        for i in range(total):
            lib.oo_polyblep_process(C.byref(o))
            out[i] = o.output
        return out

    worst, r = _run_nodes(g, setup, ref, n=n)
    assert np.max(np.abs(r)) > 0.5
    observed.note(worst)
    assert worst <= 1e-5, worst


def test_static_simple_graph_shape():
    # benches/static_vs_runtime.rs:5-18 with an explicit output: Oscillator::sine -> TptFilter -> Gain
    import ctypes as C
    lib = ol.load()
    n = 16
    freqs = np.geomspace(55.0, 7040.0, n).astype(np.float32)
    for wave, ctor in ((ol.WAVE_SINE, "sine"), (ol.WAVE_SQUARE, "square"), (ol.WAVE_SAW, "saw")):
        g = oscen_amd.Graph("static_simple_" + ctor)
        g.input_value("frequency", 440.0, per_voice=True)
        g.output_stream("out")
        g.node("osc", "Oscillator::" + ctor, 440.0, 1.0)
        g.node("filter", "TptFilter::new", 1000.0, 0.7)
        g.node("gain", "Gain::new", 0.5)
        g.connect("frequency", "osc.frequency").connect("osc.output", "filter.input")
        g.connect("filter.output", "gain.input").connect("gain.output", "out")

        def ref(v, total):
            s = ol.StaticSimple()
            lib.oo_static_simple_new(C.byref(s))
            s.osc.waveform = wave
            lib.oo_static_simple_init(C.byref(s), 44100.0)
            s.osc.frequency = float(freqs[v])
            out = np.zeros(total, dtype=np.float32)
            for i in range(total):
                lib.oo_static_simple_process(C.byref(s))
                out[i] = s.gain.output
            return out

        worst, r = _run_nodes(g, lambda e: e.set_voice_values("frequency", freqs), ref, n=n, sr=44100.0, blocks=2)
        observed.note(worst)
        assert worst <= 1e-5, (ctor, worst)


def test_static_complex_graph_shape():
    # benches/static_vs_runtime.rs:21-66 (wired path osc1->mix1->mixer->filter->vca, env -> f_mod), gated here
    import ctypes as C
    lib = ol.load()
    n = 16
    g = oscen_amd.Graph("static_complex")
    g.input_event("gate")
    g.output_stream("out")
    g.node("osc1", "PolyBlepOscillator::saw", 440.0, 0.33)
    g.node("mix1", "Gain::new", 1.0)
    g.node("mixer", "Gain::new", 1.0)
    g.node("filter_env", "AdsrEnvelope::new", 0.01, 0.3, 0.5, 0.2)
    g.node("env_amount", "Gain::new", 2000.0)
    g.node("filter", "TptFilter::new", 800.0, 0.7)
    g.node("amp_env", "AdsrEnvelope::new", 0.01, 0.2, 0.7, 0.3)
    g.node("vca", "Gain::new", 1.0)
    for s, d in [("gate", "filter_env.gate"), ("gate", "amp_env.gate"), ("osc1.output", "mix1.input"),
                 ("mix1.output", "mixer.input"), ("mixer.output", "filter.input"),
                 ("filter_env.output", "env_amount.input"), ("env_amount.output", "filter.f_mod"),
                 ("filter.output", "vca.input"), ("amp_env.output", "vca.gain"), ("vca.output", "out")]:
        g.connect(s, d)
    eng = oscen_amd.Engine(g, n, sample_rate=44100.0)
    eng.set_voice_taps(list(range(n)))
    refs = []
    for v in range(n):
        c = ol.StaticComplex()
        lib.oo_static_complex_new(C.byref(c))
        lib.oo_static_complex_init(C.byref(c), 44100.0)
        refs.append(c)
    worst = 0.0
    for b in range(4):
        frames = 256
        gates = {}
        if b == 0:
            gates = {v: (v * 3, 0.5 + 0.03 * v) for v in range(n)}
        if b == 2:
            gates = {v: (100 + v, 0.0) for v in range(n)}
        for v, (fr, val) in gates.items():
            eng.push_voice_event("gate", v, fr, val)
        eng.process_block(frames)
        got = eng.read_voice_taps(frames)
        ref = np.zeros((n, frames), dtype=np.float32)
        for v in range(n):
            c = refs[v]
            for i in range(frames):
                if v in gates and gates[v][0] == i:
                    ev = ol.Event(i, gates[v][1], 0)
                    lib.oo_adsr_handle_gate_event(C.byref(c.filter_env), C.byref(ev))
                    lib.oo_adsr_handle_gate_event(C.byref(c.amp_env), C.byref(ev))
                lib.oo_static_complex_process(C.byref(c))
                ref[v, i] = c.vca.output
        worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))))
    observed.note(worst)
    assert worst <= 1e-5, worst


@pytest.mark.parametrize("depth", [0, 2, 4])
def test_custom_graph_pipelined_variants_through_hiprtc(depth, monkeypatch):
    """A description that is not built in, large enough for the 2- and 4-wave pipelines (two oscillators,
    an envelope, two filters): every variant is compiled by hiprtc and must agree with the oracle nodes."""
    import ctypes as C
    monkeypatch.setenv("OSCEN_GPU_SPLIT", str(depth))
    lib = ol.load()
    sr = 48000.0
    text = """
    name: TwoOscVoice;
    input frequency: value = 220.0;
    input gate: event;
    input cutoff: value = 1500.0;
    output out: stream;
    nodes {
        a = PolyBlepOscillator::saw(220.0, 0.4);
        b = PolyBlepOscillator::square(220.0, 0.3);
        env = AdsrEnvelope::new(0.004, 0.03, 0.5, 0.05);
        f1 = TptFilter::new(1500.0, 1.2);
        f2 = IirLowpass::new(3000.0, 0.9);
        amp = Gain::new(0.8);
    }
    connections {
        frequency -> a.frequency;
        frequency * 1.5 -> b.frequency;
        gate -> env.gate;
        cutoff -> f1.cutoff;
        a.output + b.output -> f1.input;
        f1.output * env.output -> f2.input;
        f2.output -> amp.input;
        amp.output -> out;
    }
    """
    g = oscen_amd.Graph(dsl=text, per_voice=["frequency"])
    src = g.kernel_source()
    assert "voice_block_p2" in src and "voice_block_p4" in src
    n, frames, blocks = 70, 200, 4
    freqs = np.geomspace(55.0, 1760.0, n).astype(np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=sr)
    assert eng.pipeline_depth == max(1, depth)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    for v in range(n):
        eng.schedule_voice_event("gate", v, 3 + v % 50, 0.8)
        eng.schedule_voice_event("gate", v, 400 + v, 0.0)
    got = []
    for _ in range(blocks):
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
    got = np.concatenate(got, axis=1)

    def ref(v):
        a, b = ol.PolyBlep(), ol.PolyBlep()
        lib.oo_polyblep_new(C.byref(a), 220.0, 0.4, ol.PB_SAW)
        lib.oo_polyblep_new(C.byref(b), 220.0, 0.3, ol.PB_SQUARE)
        e = ol.Adsr()
        lib.oo_adsr_new(C.byref(e), 0.004, 0.03, 0.5, 0.05)
        f1 = ol.Tpt()
        lib.oo_tpt_new(C.byref(f1), 1500.0, 1.2, 1)
        f2 = ol.IirLowpass()
        lib.oo_iir_lowpass_new(C.byref(f2), 3000.0, 0.9)
        a.sample_rate = b.sample_rate = e.sample_rate = f1.sample_rate = f2.sample_rate = sr
        lib.oo_adsr_prepare(C.byref(e))
        lib.oo_tpt_prepare(C.byref(f1))
        lib.oo_iir_lowpass_prepare(C.byref(f2))
        a.frequency = float(freqs[v])
        b.frequency = float(np.float32(freqs[v]) * np.float32(1.5))
        out = np.zeros(frames * blocks, dtype=np.float32)
        for i in range(frames * blocks):
            if i == 3 + v % 50:
                ev = ol.Event(0, 0.8, 0)
                lib.oo_adsr_handle_gate_event(C.byref(e), C.byref(ev))
            if i == 400 + v:
                ev = ol.Event(0, 0.0, 0)
                lib.oo_adsr_handle_gate_event(C.byref(e), C.byref(ev))
            lib.oo_polyblep_process(C.byref(a))
            lib.oo_polyblep_process(C.byref(b))
            lib.oo_adsr_process(C.byref(e))
            f1.cutoff = 1500.0
            f1.input[0] = float(np.float32(a.output) + np.float32(b.output))
            lib.oo_tpt_process(C.byref(f1))
            f2.input = float(np.float32(f1.output[0]) * np.float32(e.output))
            lib.oo_iir_lowpass_process(C.byref(f2))
            out[i] = np.float32(f2.output) * np.float32(0.8)
        return out

    r = np.stack([ref(v) for v in range(n)])
    assert np.max(np.abs(r)) > 0.05
    err = float(np.max(np.abs(got - r) / np.maximum(1.0, np.abs(r))))
    observed.note(err)
    assert err <= 1e-5, err
