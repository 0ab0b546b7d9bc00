"""A graph that was NOT compiled ahead of time goes through hiprtc at og_create()."""
import numpy as np
import pytest

import oscen_amd
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
    assert worst <= 1e-5, worst
