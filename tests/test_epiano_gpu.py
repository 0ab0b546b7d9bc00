"""Electric-piano voice bank (BASELINE.json configs[2], the reference's real graph): AmplitudeSource ->
OscillatorBank per voice (32 harmonics on 32 lanes), summed, then the stereo Tremolo on the bus."""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
SR = 48000.0


def test_epiano_bank_parity_and_stereo_bus():
    n = 70
    rng = np.random.default_rng(31)
    freqs = oscen_amd.midi_note_to_freq(rng.integers(36, 97, n)).astype(np.float32)
    eng = oscen_amd.Engine("epiano_voice", n, sample_rate=SR)
    assert eng.channels == 2
    bank = ol.Bank(ol.BANK_EPIANO, n, SR)
    eng.set_voice_values("frequency", freqs)
    for v in range(n):
        bank.set_voice_frequency(v, float(freqs[v]))
    eng.set_voice_taps(list(range(n)))
    on = rng.integers(0, 200, n)
    off = on + rng.integers(1500, 4000, n)
    re = off + rng.integers(300, 1500, n)
    vel = (rng.integers(32, 128, n) / 127.0).astype(np.float32)
    events = sorted([(int(on[v]), v, float(vel[v])) for v in range(n)] + [(int(off[v]), v, 0.0) for v in range(n)] +
                    [(int(re[v]), v, float(vel[v])) for v in range(n)])
    worst, worst_bus, f0 = 0.0, 0.0, 0
    for b in range(24):
        frames = 256
        if b == 8:
            eng.set_value("brightness", 55.0); bank.set_value(ol.EPIANO_PARAMS.index("brightness"), 55.0)
            eng.set_value("vibrato_speed", 6.5); bank.set_value(ol.EPIANO_PARAMS.index("vibrato_speed"), 6.5)
        for fr, v, val in events:
            if f0 <= fr < f0 + frames:
                eng.push_voice_event("gate", v, fr - f0, val)
                bank.push_event(v, fr - f0, ol.EV_GATE, val)
        if b == 12:  # a frequency change re-derives the rotation multipliers (cos/sin) on the device
            eng.push_voice_value("frequency", 3, 17, 523.25); bank.push_event(3, 17, ol.EV_FREQ, 523.25)
        bus = eng.process_block(frames)
        taps = eng.read_voice_taps(frames)
        ref_bus, ref_taps = bank.process_block(frames, taps=list(range(n)))
        worst = max(worst, float(np.max(np.abs(taps - ref_taps) / np.maximum(1.0, np.abs(ref_taps)))))
        scale = max(1.0, float(np.max(np.sum(np.abs(ref_taps), axis=0))))
        worst_bus = max(worst_bus, float(np.max(np.abs(bus - ref_bus))) / scale)
        f0 += frames
    assert np.max(np.abs(ref_taps)) > 1e-3
    observed.note(worst)
    assert worst <= 1e-5, worst
    assert worst_bus <= 1e-5, worst_bus
    assert np.max(np.abs(bus[:, 0] - bus[:, 1])) > 1e-4  # the tremolo really pans
