"""Host-side MIDI front end (MidiParser -> VoiceAllocator -> MidiVoiceHandler): the reference's own
unit tests restated against og_midi (oscen-lib/src/voice_allocator.rs:156-258, midi.rs:237-275). CPU only."""
import numpy as np

import oscen_amd


def alloc(m, note):
    m.note_on(note)
    m.flush()
    return m.pop_outputs()[-1][0]


def test_voice_allocation_and_stealing():
    m = oscen_amd.Midi(n_voices=4)
    assert [alloc(m, n) for n in (60, 64, 67, 72)] == [0, 1, 2, 3]       # test_voice_allocation
    assert all(m.voice_state(v)["active"] for v in range(4))
    assert alloc(m, 76) == 0 and m.voice_state(0)["note"] == 76          # test_voice_stealing: oldest


def test_find_and_release_voice():
    m = oscen_amd.Midi(n_voices=4)
    alloc(m, 60); alloc(m, 64)
    m.note_off(64); m.flush()
    out = m.pop_outputs()
    assert out == [(1, 0, None, 0.0)]                                    # gate-off reaches voice 1
    s = m.voice_state(1)
    assert s["active"] and s["released"] and s["note"] is None
    m.note_off(64); m.flush()
    assert m.pop_outputs() == []                                         # not found any more


def test_prefer_released_voices_for_stealing():
    m = oscen_amd.Midi(n_voices=4)
    for n in (60, 64, 67, 72):
        alloc(m, n)
    m.note_off(64); m.flush(); m.pop_outputs()
    assert alloc(m, 76) == 1
    assert m.voice_state(1)["note"] == 76 and not m.voice_state(1)["released"]


def test_releasing_voice_continues_to_sound():
    m = oscen_amd.Midi(n_voices=2)
    assert alloc(m, 60) == 0
    m.note_off(60); m.flush(); m.pop_outputs()
    assert m.voice_state(0)["active"] and m.voice_state(0)["released"]
    assert alloc(m, 64) == 1
    assert alloc(m, 67) == 0                                             # steals the released voice


def test_parser_and_handler_contract():
    m = oscen_amd.Midi(n_voices=8)
    m.send([0x90, 69, 100], frame_offset=17)      # note on A4 vel 100
    m.send([0x90, 60, 0], frame_offset=5)         # note-on vel 0 == note off (nothing held: ignored)
    m.send([0x91, 81, 127], frame_offset=3)       # channel nibble ignored
    m.send([0xB0, 1, 2], frame_offset=0)          # CC: ignored
    m.send([0x90, 60], frame_offset=0)            # short message: ignored
    m.flush()
    out = m.pop_outputs()
    assert [(o[0], o[1]) for o in out] == [(0, 3), (1, 17)]              # applied in frame order
    assert abs(out[0][2] - 880.0) < 0.01 and out[0][3] == 1.0
    assert out[1][2] == 440.0 and abs(out[1][3] - 100.0 / 127.0) < 1e-7
    m.send([0x80, 69, 0], frame_offset=9); m.flush()
    assert m.pop_outputs() == [(1, 9, None, 0.0)]
    # 65 536-voice allocator: first free voice, then LRU
    big = oscen_amd.Midi(n_voices=65536)
    for i in range(300):
        big.note_on(30 + i % 60, frame_offset=i % 256)
    big.flush()
    voices = sorted(o[0] for o in big.pop_outputs())
    assert voices == list(range(300))


def test_wav_writer_roundtrip(tmp_path):
    import wave
    import numpy as np

    t = np.arange(4800, dtype=np.float32) / 48000.0
    stereo = np.stack([0.5 * np.sin(2 * np.pi * 440 * t), -0.25 * np.sin(2 * np.pi * 220 * t)], axis=1).astype(np.float32)
    p16 = tmp_path / "a.wav"
    oscen_amd.write_wav(p16, stereo, 48000, 16)
    with wave.open(str(p16)) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (2, 2, 48000, 4800)
        pcm = np.frombuffer(w.readframes(4800), dtype="<i2").reshape(-1, 2)
    assert np.max(np.abs(pcm / 32767.0 - stereo)) < 1.0 / 32767.0
    p32 = tmp_path / "b.wav"
    oscen_amd.write_wav(p32, stereo[:, 0], 48000, 32)
    raw = open(p32, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE" and raw[20:22] == b"\x03\x00"
    assert np.array_equal(np.frombuffer(raw[44:], dtype="<f4"), stereo[:, 0])


def test_midi_in_queue_capacity_and_late_messages():
    """`midi_in` holds 32 messages per block for up to 24 voices (graph/types.rs:18): the 33rd try_push fails and the
    message is dropped; a message whose frame_offset lies beyond the block never reaches the allocator."""
    m = oscen_amd.Midi(n_voices=8)
    rcs = [m.lib.og_midi_send(m.h, bytes([0x90, 40 + i, 100]), 3, i) for i in range(40)]
    assert rcs[:32] == [0] * 32 and rcs[32:] == [oscen_amd.OG_E_OVERFLOW] * 8
    assert m.dropped == 8
    m.flush()
    assert len(m.pop_outputs()) == 32
    m.set_queue_capacity(100)
    assert m.send_many(np.arange(50, 90, dtype=np.uint8), np.arange(40), on=True) == 0
    # the allocator is O(log N): same decisions as the reference's scans (first free, then released-first LRU)
    big = oscen_amd.Midi(n_voices=100000)
    big.set_queue_capacity(1 << 20)
    notes = (np.arange(200000) % 128).astype(np.uint8)
    assert big.send_many(notes, np.zeros(200000, dtype=np.uint32), on=True) == 0
    big.flush()
    st = big.voice_state(0)
    assert st["active"] and st["age"] == 100000  # voice 0 was the oldest held voice when the 100001st note arrived


class ScanAllocator:
    """voice_allocator.rs:57-108 as written: a scan over all voices per decision (what og_midi's indexes must reproduce)"""

    def __init__(self, n):
        self.v = [dict(active=False, released=False, note=None, age=0) for _ in range(n)]
        self.age = 0

    def allocate(self, note):
        for i, s in enumerate(self.v):                                    # :59-68 first inactive voice
            if not s["active"]:
                break
        else:                                                             # :72-80 (released first, then oldest)
            i = min(range(len(self.v)), key=lambda k: (0 if self.v[k]["released"] else 1, self.v[k]["age"]))
        self.v[i] = dict(active=True, released=False, note=note, age=self.age)
        self.age += 1
        return i

    def note_off(self, note):                                             # :92-108
        for i, s in enumerate(self.v):
            if s["active"] and not s["released"] and s["note"] == note:
                s["released"], s["note"] = True, None
                return i
        return None


def test_allocator_indexes_equal_the_reference_scan_under_heavy_stealing():
    """og_midi answers allocate / find / release from fixed-size indexes (held FIFO, released heap, per-note heaps
    with positions) instead of the reference's scans: every decision and the whole voice table must still equal the
    scan's, through thousands of steals, repeated notes on several voices and note-offs for notes nobody holds"""
    rng = np.random.default_rng(20260927)
    for n, n_notes, p_off, steps in ((1, 3, 0.3, 300), (5, 4, 0.45, 3000), (24, 12, 0.5, 6000), (67, 128, 0.35, 8000), (300, 6, 0.55, 12000)):
        m = oscen_amd.Midi(n_voices=n)
        m.set_queue_capacity(1 << 20)
        ref = ScanAllocator(n)
        want = []
        for step in range(steps):
            note = int(rng.integers(30, 30 + n_notes))
            if rng.random() < p_off:
                m.note_off(note)
                i = ref.note_off(note)
                if i is not None:
                    want.append((i, False))
            else:
                m.note_on(note, velocity=100)
                want.append((ref.allocate(note), True))
            if step % 97 == 96 or step == steps - 1:
                m.flush()
                got = [(voice, hz is not None) for voice, _fo, hz, _gate in m.pop_outputs()]
                assert got == want, (n, step)
                want = []
                for k in range(n):
                    s = m.voice_state(k)
                    r = ref.v[k]
                    assert (s["active"], s["released"], s["note"]) == (r["active"], r["released"], r["note"]), (n, step, k)
                    if r["active"]:
                        assert s["age"] == r["age"]
