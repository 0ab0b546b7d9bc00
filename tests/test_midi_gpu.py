"""MIDI bytes -> allocator -> voice bank on the GPU, against the oracle fed with the same routed events."""
import numpy as np
import pytest

import oscen_amd
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_midi_driven_bank_matches_oracle():
    n, frames, sr = 8, 256, 48000.0
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=sr)
    eng.set_voice_taps(list(range(n)))
    midi = oscen_amd.Midi(eng)
    twin = oscen_amd.Midi(n_voices=n)          # same allocator, detached: tells us what each voice receives
    bank = ol.Bank(ol.BANK_FM, n, sr)
    rng = np.random.default_rng(5)
    held = []
    worst = 0.0
    for b in range(20):
        for _ in range(rng.integers(0, 4)):
            fo = int(rng.integers(0, frames))
            if held and rng.random() < 0.45:
                note = held.pop(int(rng.integers(0, len(held))))
                msg = [0x80, note, 0]
            else:
                note = int(rng.integers(40, 90))
                held.append(note)
                msg = [0x90, note, int(rng.integers(30, 128))]
            midi.send(msg, fo)
            twin.send(msg, fo)
        twin.flush()
        for voice, fo, hz, gate in twin.pop_outputs():
            if hz is not None:
                bank.push_event(voice, fo, ol.EV_FREQ, hz)
            bank.push_event(voice, fo, ol.EV_GATE, gate)
        bus = midi.process_block(frames)
        taps = eng.read_voice_taps(frames)
        ref_bus, ref = bank.process_block(frames, taps=list(range(n)))
        worst = max(worst, float(np.max(np.abs(taps - ref) / np.maximum(1.0, np.abs(ref)))))
    assert np.max(np.abs(ref)) > 0.01
    assert worst <= 1e-5, worst
