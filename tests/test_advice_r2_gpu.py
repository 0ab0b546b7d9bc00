"""Regression tests for the round-2 advisor findings, all through the C ABI:

* the blocking drop-in entry og_process_block with the DEFAULT batching (one block per launch) must see its
  completion word -- it used to wait for a marker that was never launched and fell into the 20 ms fallback;
* og_save_state with blocks still queued must not save events those blocks consume;
* a try_push'ed event whose frame_offset >= frames is dropped even when a flush (a setter, a blocking block after
  async ones) happens between the push and the block;
* og_load_state rejects blobs whose event records address voices / inputs the graph does not have.
"""
import ctypes as C
import time

import numpy as np
import pytest

import oscen_amd

pytestmark = pytest.mark.gpu

SR = 48000.0


def blocking_stats(eng):
    calls, timeouts = C.c_uint64(), C.c_uint64()
    assert eng.lib.og_blocking_stats(eng.h, C.byref(calls), C.byref(timeouts)) == 0
    return calls.value, timeouts.value


@pytest.mark.parametrize("batching", [None, 1, 8])
def test_blocking_block_sees_its_completion_word(batching):
    """og_process_block at bus_batch = 1 (the default), 1 set explicitly and 8: every call ends on the marker word,
    none in the 20 ms fallback; a 256-frame block of 4 096 voices takes far less than a millisecond of wall time"""
    n = 4096
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    if batching is not None:
        eng.set_bus_batching(batching)
    plans = oscen_amd.note_plans(n)
    eng.set_voice_values("frequency", plans["frequency"])
    for v in range(0, n, 3):
        eng.push_voice_event("gate", v, v % 256, 0.8)
    for _ in range(5):
        eng.process_block(256)
    lat = []
    for _ in range(60):
        t0 = time.perf_counter()
        bus = eng.process_block(256)
        lat.append(time.perf_counter() - t0)
    assert np.abs(bus).max() > 0.0
    calls, timeouts = blocking_stats(eng)
    assert calls == 65 and timeouts == 0, (calls, timeouts)
    assert np.median(lat) < 2e-3, np.median(lat)


def test_midi_blocking_block_sees_its_completion_word():
    n = 2048
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    midi = oscen_amd.Midi(eng)
    for b in range(20):
        midi.note_on(40 + b, 100, frame_offset=b)
        out = midi.process_block(256)
    assert np.abs(out).max() > 0.0
    calls, timeouts = blocking_stats(eng)
    assert calls == 20 and timeouts == 0, (calls, timeouts)


def test_snapshot_with_queued_blocks_does_not_replay_their_events():
    """save_state while og_set_bus_batching(8) still holds blocks whose frames contain scheduled events: the blob's
    frame counter is past them, so they must not be in it (they used to fire a second time after the load)"""
    n, block = 128, 128
    plans = oscen_amd.note_plans(n)

    def fresh(batch):
        e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        e.set_bus_batching(batch)
        e.set_voice_values("frequency", plans["frequency"])
        for v in range(n):
            e.schedule_voice_event("gate", v, 5 + v, 0.9)           # inside the queued blocks
            e.schedule_voice_event("gate", v, 300 + 2 * v, 0.0)     # inside the queued blocks (v < 42), later for the rest
            e.schedule_voice_event("gate", v, 900 + v, 0.7)         # after the snapshot point
        return e

    eng = fresh(8)
    for _ in range(3):
        eng.process_block_async(block)          # queued, not launched: 3 < 8
    n_bytes = eng.state_bytes                   # (counts with the same horizon the save uses)
    blob = eng.save_state()
    assert blob.nbytes == n_bytes
    want = eng.render(12 * block, block=block)

    ref = fresh(1)                              # block by block, no queue: the truth
    for _ in range(3):
        ref.process_block(block)
    truth = ref.render(12 * block, block=block)
    assert np.array_equal(want, truth)

    other = fresh(8)
    other.load_state(blob)
    assert other.frames_processed == 3 * block
    got = other.render(12 * block, block=block)
    assert np.array_equal(got, truth)


def test_late_try_push_is_dropped_even_when_a_flush_intervenes():
    """og_push_voice_event(frame_offset >= frames of the next block) is never delivered (the reference clears its
    queues at the end of the block) -- also when async blocks are queued and a setter / a blocking block launches
    them between the push and the block"""
    n = 64
    plans = oscen_amd.note_plans(n)

    def run(with_queue):
        e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        e.set_voice_values("frequency", plans["frequency"])
        if with_queue:  # (no taps yet: a tapped engine launches every block at once)
            e.set_bus_batching(8)
            e.process_block_async(128)
            e.process_block_async(128)
        else:
            e.process_block(128)
            e.process_block(128)
        assert e.push_voice_event("gate", 3, 10, 0.9) == 0       # delivered: frame 10 of the next block
        assert e.push_voice_event("gate", 5, 200, 0.9) == 0      # frame_offset 200 >= 128: dropped with the block
        e.set_value("op2_level", 0.4)                            # a setter launches the queue
        e.set_voice_taps(np.arange(n, dtype=np.uint32))
        a = e.process_block(128)
        ta = e.read_voice_taps(128)
        b = e.process_block(128)
        tb = e.read_voice_taps(128)
        return e.events_dropped, a, b, ta, tb

    d0, a0, b0, ta0, tb0 = run(False)
    d1, a1, b1, ta1, tb1 = run(True)
    assert d0 == 1 and d1 == 1, (d0, d1)
    assert np.abs(ta0[3]).max() > 0.0 and np.abs(ta0[5]).max() == 0.0 and np.abs(tb0[5]).max() == 0.0
    assert np.array_equal(ta0, ta1) and np.array_equal(tb0, tb1)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)


def test_load_state_rejects_crafted_event_records():
    n = 32
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    eng.set_voice_values("frequency", oscen_amd.note_plans(n)["frequency"])
    eng.schedule_voice_event("gate", 1, 1000, 0.5)
    eng.schedule_voice_event("gate", 2, 2000, 0.0)
    eng.process_block(64)
    blob = eng.save_state()
    eng.load_state(blob)  # the honest blob loads
    # layout of the control block (og_engine.cpp): header {magic, version, frame_now u64, n_inputs, active_ramps,
    # n_events u64}, values, ramps, then 24-byte event records {voice, target, frame u64, value, block_local}
    n_inputs = eng.lib.og_num_inputs(eng.h)
    ctrl = blob.nbytes - (32 + n_inputs * (4 + 16) + 2 * 24)
    hdr = np.frombuffer(blob[ctrl:ctrl + 32].tobytes(), dtype=np.uint32)
    assert hdr[0] == 0x3253474F and hdr[4] == n_inputs and hdr[6] == 2
    ev0 = ctrl + 32 + n_inputs * 20

    def patched(off, value, dtype):
        b = blob.copy()
        b[off:off + np.dtype(dtype).itemsize] = np.frombuffer(np.array([value], dtype=dtype).tobytes(), dtype=np.uint8)
        return b

    for bad in (patched(ev0 + 4, 77, np.uint32),                       # event input 77 does not exist
                patched(ev0 + 4, 0x80000000 | 1, np.uint32),           # SETVALUE on a broadcast input
                patched(ev0, n + 5, np.uint32),                        # voice out of range
                patched(ctrl + 24, (1 << 61) + 2, np.uint64)):         # n_events that wraps the size arithmetic
        with pytest.raises(oscen_amd.OscenError):
            eng.load_state(bad)
    eng.load_state(blob)
    assert eng.process_block(64).shape == (64, 1)
