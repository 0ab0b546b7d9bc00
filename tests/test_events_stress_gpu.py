"""The live event path under stress: the same score reaches three engines in three ways -- (A) scheduled in bulk up
front (one timeline rebuild), (B) pushed block by block with try_push semantics plus events scheduled a few blocks
ahead (incremental per-voice segments, with the head-room shrunk so that the event ring wraps many times), (C) as
(B) on the asynchronous entry with the block queue on (pushes force queue launches).  All three must agree bit for bit
with each other (state snapshot and final bus) and with the CPU oracle."""
import os

import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
SR = 48000.0


def make_score(n, blocks, block, seed):
    rng = np.random.default_rng(seed)
    k = 1 if n < 1000 else 6  # (more than V/2 new events in one block take the bulk path: keep the big bank below that)
    ev = []  # (abs frame, voice, value)
    for v in range(n):
        t = int(rng.integers(0, 300))
        while t < blocks * block:
            ev.append((t, v, float(np.float32(rng.integers(32, 128)) / np.float32(127.0))))
            t += k * int(rng.integers(40, 700))
            if t >= blocks * block:
                break
            ev.append((t, v, 0.0))
            t += k * int(rng.integers(1, 400))
    ev.sort(key=lambda e: (e[0], e[1]))
    return ev


@pytest.mark.parametrize("n", [130, 5000])
def test_bulk_incremental_and_queued_event_paths_agree(n, monkeypatch):
    blocks, block = 40, 192
    score = make_score(n, blocks, block, seed=int(os.environ.get("OSCEN_STRESS_SEED", n)))  # (soak runs: other scores)
    freqs = oscen_amd.midi_note_to_freq(np.random.default_rng(3).integers(36, 97, n)).astype(np.float32)
    probe = np.unique(np.linspace(0, n - 1, 24).astype(np.uint32))

    def engine():
        e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        e.set_voice_values("frequency", freqs)
        return e

    # (A) bulk
    a = engine()
    a.schedule_voice_events("gate", [v for _, v, _ in score], [f for f, _, _ in score], [x for _, _, x in score])
    a.set_voice_taps(probe)
    taps_a = []
    for _ in range(blocks):
        a.process_block(block)
        taps_a.append(a.read_voice_taps(block))
    taps_a = np.concatenate(taps_a, axis=1)
    bus_a = a.process_block(64)
    state_a = a.save_state()
    assert a.event_stats["incremental_updates"] == 0

    def live(e, queued):
        """events of block b: those in its first half pushed block-locally right before it, those in its second half
        scheduled two blocks earlier (so that unconsumed old events and new ones get merged)"""
        if queued:
            e.set_bus_batching(8)
        by_block = [[] for _ in range(blocks)]
        for f, v, x in score:
            by_block[f // block].append((f, v, x))
        for b in range(blocks):
            for k in (b, b + 1, b + 2):  # schedule ahead the late halves of blocks b+2 (and, at the start, of b, b+1)
                if k >= blocks or (k != b + 2 and b != 0):
                    continue
                for f, v, x in by_block[k]:
                    if f % block >= block // 2:
                        e.schedule_voice_event("gate", v, f, x)
            for f, v, x in by_block[b]:
                if f % block < block // 2:
                    assert e.push_voice_event("gate", v, f - b * block, x) == 0
            if queued:
                e.process_block_async(block)
            else:
                e.process_block(block)
        return e.process_block(64), e.save_state()

    monkeypatch.setenv("OSCEN_GPU_EXPERIMENTAL", "1")
    monkeypatch.setenv("OSCEN_GPU_EV_HEADROOM", "4096")  # a small ring: many wraps (and merges of old + new events)
    bmode = engine()
    bus_b, state_b = live(bmode, queued=False)
    stats = bmode.event_stats
    # the device event buffer is a ring: with the head-room shrunk the append pointer wraps into the space of consumed /
    # superseded segments instead of rebuilding the whole timeline
    assert stats["incremental_updates"] > blocks // 2 and stats["full_rebuilds"] >= 1, stats
    assert stats["ring_wraps"] >= (1 if n > 1000 else 0), stats
    c = engine()
    bus_c, state_c = live(c, queued=True)
    monkeypatch.delenv("OSCEN_GPU_EV_HEADROOM")

    def dsp(blob):  # the DSP part of a snapshot (the control block lists pending events in path-dependent order)
        return blob[: a.state_words_per_voice * n * 4]

    assert np.array_equal(bus_a, bus_b) and np.array_equal(bus_a, bus_c)
    assert np.array_equal(dsp(state_a), dsp(state_b)) and np.array_equal(dsp(state_a), dsp(state_c))

    # and against the oracle (the probed voices in a small bank)
    bank = ol.Bank(ol.BANK_FM, len(probe), SR)
    idx = {int(v): i for i, v in enumerate(probe)}
    for v, i in idx.items():
        bank.set_voice_frequency(i, float(freqs[v]))
    ref = []
    for b in range(blocks):
        for f, v, x in score:
            if v in idx and b * block <= f < (b + 1) * block:
                bank.push_event(idx[v], f - b * block, ol.EV_GATE, x)
        _, t = bank.process_block(block, taps=list(range(len(probe))))
        ref.append(t)
    ref = np.concatenate(ref, axis=1)
    err = float(np.max(np.abs(taps_a - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5 and np.abs(ref).max() > 1e-2, err
