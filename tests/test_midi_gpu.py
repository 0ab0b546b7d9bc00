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


def test_midi_into_a_cluster_matches_midi_into_one_engine():
    """og_midi_create_cluster: one allocator over the global voice ids of a 3-shard bank; the same MIDI stream into a
    single engine of the same size gives the same voices (taps bit for bit) and the same bus up to the association of
    the cross-shard sum"""
    n, frames, sr = 96, 256, 48000.0
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=sr)
    cl = oscen_amd.Cluster("fm_voice", n, [0, 0, 0], sample_rate=sr)
    m_eng, m_cl = oscen_amd.Midi(eng), oscen_amd.Midi(cl)
    for m in (m_eng, m_cl):
        m.set_queue_capacity(256)
    shards = [cl.shard(s) for s in range(3)]
    for sh in shards:
        sh.set_voice_taps(np.arange(sh.n_voices, dtype=np.uint32))
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    rng = np.random.default_rng(11)
    held = []
    for b in range(24):
        for _ in range(int(rng.integers(0, 40))):    # enough notes to run the allocator through all three shards and into stealing
            fo = int(rng.integers(0, frames))
            if held and rng.random() < 0.4:
                msg = [0x80, held.pop(int(rng.integers(0, len(held)))), 0]
            else:
                note = int(rng.integers(36, 97))
                held.append(note)
                msg = [0x90, note, int(rng.integers(30, 128))]
            m_eng.send(msg, fo)
            m_cl.send(msg, fo)
        a = m_eng.process_block(frames)
        t_eng = eng.read_voice_taps(frames)
        c = m_cl.process_block(frames)[:, 0]
        t_cl = np.concatenate([sh.read_voice_taps(frames) for sh in shards], axis=0)
        assert np.array_equal(t_eng, t_cl)
        assert np.max(np.abs(a[:, 0] - c)) <= 1e-5 * max(1.0, float(np.abs(a).max()))
    assert np.abs(a).max() > 1e-2
    for v in (0, 40, 95):
        assert m_eng.voice_state(v) == m_cl.voice_state(v)
    assert cl.lib.og_midi_process_block_async(m_cl.h, frames, None) == oscen_amd.OG_E_UNSUPPORTED
