"""GPU parity tests: the HIP voice kernels (through the C ABI) against the CPU
oracle on identical note events.  Tolerance per BASELINE.json north_star:
|a - b| <= 1e-5 * max(1, |ref|) on every per-voice sample; the mix bus is
judged against the oracle's f64 sum of the per-voice outputs (the reference's
sequential f32 fold and the GPU's fixed tree differ only by re-association).
"""
import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

TOL = 1e-5
SR = 48000.0


def rel_err(got, ref):
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))


class Pair:
    """An engine and an oracle bank driven by the same calls."""

    def __init__(self, graph, kind, n, params, sr=SR):
        self.eng = oscen_amd.Engine(graph, n, sample_rate=sr)
        self.bank = ol.Bank(kind, n, sr)
        self.params = params
        self.n = n
        self.taps = list(range(n))
        self.eng.set_voice_taps(self.taps)

    def set_value(self, name, v):
        self.eng.set_value(name, v)
        self.bank.set_value(self.params.index(name), v)

    def set_value_with_ramp(self, name, v, frames):
        self.eng.set_value_with_ramp(name, v, frames)
        self.bank.set_value_with_ramp(self.params.index(name), v, frames)

    def set_value_immediate(self, name, v):
        self.eng.set_value_immediate(name, v)
        self.bank.set_value_immediate(self.params.index(name), v)

    def set_freqs(self, freqs):
        self.eng.set_voice_values("frequency", np.asarray(freqs, dtype=np.float32))
        for v, f in enumerate(freqs):
            self.bank.set_voice_frequency(v, float(f))

    def gate(self, voice, off, vel):
        assert self.eng.push_voice_event("gate", voice, off, vel) == 0
        self.bank.push_event(voice, off, ol.EV_GATE, vel)

    def freq(self, voice, off, hz):
        self.eng.push_voice_value("frequency", voice, off, hz)
        self.bank.push_event(voice, off, ol.EV_FREQ, hz)

    def block(self, frames):
        bus = self.eng.process_block(frames)
        taps = self.eng.read_voice_taps(frames)
        ref_bus, ref_taps = self.bank.process_block(frames, taps=self.taps)
        return bus, taps, ref_bus, ref_taps, self.bank.last_bus_f64(frames)


def random_note_script(n, total, seed, span=(300, 1500)):
    """per-voice (on, off, retrig) frames inside `total`, plus velocities"""
    rng = np.random.default_rng(seed)
    on = rng.integers(0, 256, n)
    off = on + rng.integers(span[0], span[1], n)
    re = off + rng.integers(100, 900, n)
    vel = rng.integers(32, 128, n).astype(np.float32) / np.float32(127.0)
    events = []
    for v in range(n):
        events += [(int(on[v]), v, float(vel[v])), (int(off[v]), v, 0.0)]
        if re[v] < total:
            events.append((int(re[v]), v, float(vel[v])))
    return sorted(events)


def run_script(pair, events, blocks, check_bus=True):
    f0 = 0
    worst = 0.0
    all_taps, all_ref = [], []
    for frames in blocks:
        for fr, v, val in events:
            if f0 <= fr < f0 + frames:
                pair.gate(v, fr - f0, val)
        bus, taps, ref_bus, ref_taps, ref64 = pair.block(frames)
        worst = max(worst, rel_err(taps, ref_taps))
        if check_bus:
            # bus: f64 oracle sum, tolerance scaled by sum |x_i| (re-association error bound)
            scale = max(1.0, float(np.max(np.sum(np.abs(ref_taps), axis=0))))
            assert np.max(np.abs(bus[:, 0] - ref64)) <= TOL * scale
        all_taps.append(taps)
        all_ref.append(ref_taps)
        f0 += frames
    return worst, np.concatenate(all_taps, axis=1), np.concatenate(all_ref, axis=1)


def midi_freqs(n, seed):
    rng = np.random.default_rng(seed)
    return oscen_amd.midi_note_to_freq(rng.integers(36, 97, n)).astype(np.float32)


# --------------------------------------------------------------------------
def test_fm_voice_config1_one_voice_one_second():
    # BASELINE.json configs[0]: 1 voice, 48 kHz, 1 s, note 69 on @0 vel 100/127, off @24000
    p = Pair("fm_voice", ol.BANK_FM, 1, ol.FM_PARAMS)
    p.set_freqs([oscen_amd.midi_note_to_freq(69)])
    events = [(0, 0, float(oscen_amd.midi_velocity_to_gate(100))), (24000, 0, 0.0)]
    worst, got, ref = run_script(p, events, [256] * 187 + [128])
    assert got.shape == (1, 48000)
    assert np.max(np.abs(ref)) > 0.05
    observed.note(worst)
    assert worst <= TOL, worst


def test_fm_bank_default_params_ragged_voice_count():
    n = 150  # not a multiple of 64: exercises the masked tail wave
    p = Pair("fm_voice", ol.BANK_FM, n, ol.FM_PARAMS)
    p.set_freqs(midi_freqs(n, 1))
    events = random_note_script(n, 4096, 2)
    worst, got, ref = run_script(p, events, [256] * 16)
    assert np.max(np.abs(ref)) > 0.05
    observed.note(worst)
    assert worst <= TOL, worst


@pytest.mark.parametrize("depth", [0, 2, 4, "4w"])
def test_fm_bank_every_pipeline_depth(depth, monkeypatch):
    """The same bank through the ordinary kernel, the 2-wave pipeline and both forms of the 4-wave pipeline -- 8-frame
    hand-offs (og_k4_*) and 16-frame ones (og_k4w_*, round 5) -- (OSCEN_GPU_SPLIT / OSCEN_GPU_WIDE pin the variant the
    engine would otherwise pick from the bank size): ragged voice count, block lengths that are not multiples of the
    hand-off or the 16-frame bus tile, ramps, per-voice frequency events."""
    wide = depth == "4w"
    depth = 4 if wide else depth
    monkeypatch.setenv("OSCEN_GPU_SPLIT", str(depth))
    monkeypatch.setenv("OSCEN_GPU_WIDE", "1" if wide else "0")
    n = 150
    p = Pair("fm_voice", ol.BANK_FM, n, ol.FM_PARAMS)
    assert p.eng.pipeline_depth == max(1, depth)
    assert p.eng.kernel_variant.startswith({0: "og_k_", 2: "og_k2_", 4: "og_k4w_" if wide else "og_k4_"}[depth])
    for op in ("op3", "op2", "op1", "filter"):
        p.set_value(op + "_attack", 0.002)
        p.set_value(op + "_decay", 0.004)
        p.set_value(op + "_release", 0.006)
    p.set_value("op3_level", 0.4)
    p.set_value("op3_feedback", 0.3)
    p.set_value("route", 0.35)
    p.set_value("filter_env_amount", 2500.0)
    p.set_freqs(midi_freqs(n, 11))
    events = random_note_script(n, 2300, 12, span=(300, 900))
    blocks = [256, 7, 249, 100, 512, 1, 15, 17, 256, 300, 333, 254]
    f0, worst = 0, 0.0
    for bi, frames in enumerate(blocks):
        for fr, v, val in events:
            if f0 <= fr < f0 + frames:
                p.gate(v, fr - f0, val)
        if bi == 3:
            p.set_value_with_ramp("filter_cutoff", 5200.0, 400)
        if bi == 5:
            for v in range(0, n, 7):
                p.freq(v, 0, 330.0 + v)
        bus, taps, ref_bus, ref_taps, ref64 = p.block(frames)
        worst = max(worst, rel_err(taps, ref_taps))
        scale = max(1.0, float(np.max(np.sum(np.abs(ref_taps), axis=0))))
        assert np.max(np.abs(bus[:, 0] - ref64)) <= TOL * scale
        f0 += frames
    observed.note(worst)
    assert worst <= TOL, worst


def test_fm_bank_fast_envelopes_all_stages():
    # short A/D/R so attack, decay, sustain, release, idle and retrigger-from-release all occur
    n = 128
    p = Pair("fm_voice", ol.BANK_FM, n, ol.FM_PARAMS)
    for op in ("op3", "op2", "op1", "filter"):
        p.set_value(op + "_attack", 0.002)
        p.set_value(op + "_decay", 0.004)
        p.set_value(op + "_release", 0.006)
    p.set_freqs(midi_freqs(n, 3))
    events = random_note_script(n, 3072, 4, span=(500, 1200))
    worst, got, ref = run_script(p, events, [256] * 12)
    observed.note(worst)
    assert worst <= TOL, worst


def test_fm_bank_variant_feedback_route_envamount_and_ramps():
    # SURVEY 8d config-2 variant: feedback, route, filter_env_amount (per-sample tan) and a cutoff ramp
    n = 96
    p = Pair("fm_voice", ol.BANK_FM, n, ol.FM_PARAMS)
    p.set_value_immediate("op3_feedback", 0.3)
    p.set_value_immediate("op2_feedback", 0.2)
    p.set_value_immediate("route", 0.5)
    p.set_value_immediate("filter_env_amount", 2000.0)
    p.set_freqs(midi_freqs(n, 5))
    events = random_note_script(n, 4096, 6)
    f0, worst = 0, 0.0
    for b in range(16):
        if b == 3:
            p.set_value("filter_cutoff", 6000.0)          # default ramp 2205 frames
            p.set_value_with_ramp("route", 0.1, 300)
        if b == 9:
            # keep the operator self-feedback loop gain 2*pi*feedback*level*env below 1:
            # beyond it phase-feedback FM is chaotic and ulp-level libm differences grow
            # without bound in ANY two implementations (not a parity regime)
            p.set_value_immediate("op3_level", 0.4)
            p.set_value("filter_resonance", 2.5)
        for fr, v, val in events:
            if f0 <= fr < f0 + 256:
                p.gate(v, fr - f0, val)
        bus, taps, ref_bus, ref_taps, ref64 = p.block(256)
        worst = max(worst, rel_err(taps, ref_taps))
        f0 += 256
    observed.note(worst)
    assert worst <= TOL, worst
    assert abs(p.eng.get_value("filter_cutoff") - 6000.0) < 1e-3


def test_block_size_does_not_change_results_bit_exact():
    # block_processing_test.rs law on the GPU: 256 vs 512 vs ragged chunks, bit-exact
    n = 70
    outs = []
    for blocks in ([256] * 8, [512] * 4, [100, 412, 1, 511, 256, 256, 512]):
        eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        eng.set_voice_values("frequency", midi_freqs(n, 7))
        eng.set_voice_taps(list(range(n)))
        eng.set_value_immediate("filter_env_amount", 1500.0)
        events = random_note_script(n, 2048, 8, span=(200, 700))
        f0, chunks, buses = 0, [], []
        for frames in blocks:
            for fr, v, val in events:
                if f0 <= fr < f0 + frames:
                    eng.push_voice_event("gate", v, fr - f0, val)
            buses.append(eng.process_block(frames))
            chunks.append(eng.read_voice_taps(frames))
            f0 += frames
        outs.append((np.concatenate(chunks, axis=1), np.concatenate(buses, axis=0)))
    for taps, bus in outs[1:]:
        assert np.array_equal(taps, outs[0][0])
    assert np.any(outs[0][0] != 0.0)


def test_event_edge_cases():
    n = 64
    p = Pair("fm_voice", ol.BANK_FM, n, ol.FM_PARAMS)
    p.set_freqs(midi_freqs(n, 9))
    # same-frame off+on, event on the last frame, frequency change with the note-on
    p.gate(0, 0, 1.0)
    p.gate(1, 255, 0.7)
    p.gate(2, 10, 0.9)
    p.gate(2, 10, 0.0)
    p.gate(2, 10, 0.5)
    p.freq(3, 100, 523.25)
    p.gate(3, 100, 0.8)
    # an event beyond the block is dropped (codegen/mod.rs:782-871), in both implementations
    p.eng.push_voice_event("gate", 4, 300, 1.0)
    p.bank.push_event(4, 300, ol.EV_GATE, 1.0)
    worst = 0.0
    for _ in range(6):
        bus, taps, ref_bus, ref_taps, _ = p.block(256)
        worst = max(worst, rel_err(taps, ref_taps))
        assert np.all(taps[4] == 0.0) and np.all(ref_taps[4] == 0.0)
    observed.note(worst)
    assert worst <= TOL, worst
    assert p.eng.events_dropped == 1
    # 33rd event on one voice/input in one block overflows like ArrayVec<_, 32>
    for i in range(32):
        assert p.eng.push_voice_event("gate", 5, i, 0.5) == 0
    assert p.eng.push_voice_event("gate", 5, 40, 0.5) == oscen_amd.OG_E_OVERFLOW
    p.eng.process_block(64)


def test_sub_voice_parity():
    # "osc+env+TptFilter" voice (perf/profile_graph.rs:12-37): PolyBLEP saw, TPT, ADSR
    n = 100
    p = Pair("sub_voice", ol.BANK_SUB, n, ol.SUB_PARAMS)
    p.set_freqs(midi_freqs(n, 11))
    events = random_note_script(n, 4096, 12)
    f0, worst = 0, 0.0
    for b in range(16):
        if b == 5:
            p.set_value("cutoff", 900.0)
            p.set_value("q", 3.0)
        for fr, v, val in events:
            if f0 <= fr < f0 + 256:
                p.gate(v, fr - f0, val)
        bus, taps, ref_bus, ref_taps, ref64 = p.block(256)
        worst = max(worst, rel_err(taps, ref_taps))
        f0 += 256
    observed.note(worst)
    assert worst <= TOL, worst


def test_state_snapshot_roundtrip():
    n = 64
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    eng.set_voice_values("frequency", midi_freqs(n, 13))
    for v in range(n):
        eng.push_voice_event("gate", v, v, 0.9)
    eng.process_block(256)
    blob = eng.save_state()
    a = eng.process_block(256)
    eng.load_state(blob)
    b = eng.process_block(256)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("graph,n", [("fm_voice", 300), ("fm_voice", 70000), ("epiano_voice", 9000), ("sat4x_voice", 66000)])
def test_batched_bus_reduce_gives_the_same_bits(graph, n):
    """og_set_bus_batching: one reduce launch per 8 blocks (per tree level; 70 000 fm voices = 1 094 partial rows and
    9 000 e-piano voices = 1 125 rows take two levels; the e-piano adds the post-mix Tremolo in block order) -- same
    tree, same association: the bus must not change by a bit, including a ragged last block and a partial last batch"""
    total, block = 256 * 10 + 100, 256
    outs = []
    for batch in (1, 8, 3):
        eng = oscen_amd.Engine(graph, n, sample_rate=SR)
        eng.set_bus_batching(batch)
        plans = oscen_amd.note_plans(n, span=total)
        oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
        outs.append(eng.render(total, block=block))
        # blocking blocks after a batched render still deliver their own bus
        assert eng.process_block(64).shape[0] == 64
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.abs(outs[0]).max() > 1e-2


def test_state_snapshot_carries_ramps_frame_counter_and_pending_events():
    """a snapshot taken in the middle of a parameter ramp, with note-offs still scheduled and a try_push'ed event queued
    for the next block, loaded into a FRESH engine: the continuation is the original render, bit for bit"""
    n, block = 96, 128
    freqs = midi_freqs(n, 5)

    def fresh():
        e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        e.set_voice_values("frequency", freqs)
        e.set_voice_taps(np.arange(n, dtype=np.uint32))
        return e

    eng = fresh()
    for v in range(n):
        eng.schedule_voice_event("gate", v, 3 + v, 0.8)
        eng.schedule_voice_event("gate", v, 700 + 5 * v, 0.0)     # still pending at the snapshot
        eng.schedule_voice_event("gate", v, 1500 + 3 * v, 0.6)
    eng.process_block(block)
    eng.set_value_with_ramp("filter_cutoff", 6000.0, 900)          # ramp spans the snapshot point
    eng.set_value("op3_feedback", 0.1)
    eng.set_value("route", 0.3)
    for _ in range(3):
        eng.process_block(block)
    assert eng.push_voice_event("gate", 7, 11, 0.0) == 0            # queued for the next block
    blob = eng.save_state()
    want = []
    for _ in range(12):
        eng.process_block(block)
        want.append(eng.read_voice_taps(block))
    other = fresh()
    other.load_state(blob)
    assert other.frames_processed == 4 * block
    got = []
    for _ in range(12):
        other.process_block(block)
        got.append(other.read_voice_taps(block))
    assert np.array_equal(np.concatenate(got, axis=1), np.concatenate(want, axis=1))
    assert abs(other.get_value("filter_cutoff") - eng.get_value("filter_cutoff")) == 0.0
    with pytest.raises(oscen_amd.OscenError):
        oscen_amd.Engine("fm_voice", n + 1, sample_rate=SR).load_state(blob)


# --------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[1]: 65 536 voices, block 256)
# --------------------------------------------------------------------------
def test_full_size_properties():
    n = 65536
    total, block = 2048, 256
    plans = oscen_amd.note_plans(n)
    sample = np.unique(np.concatenate([np.arange(0, n, 997), [63, 64, n - 1]])).astype(np.uint32)

    def run(blk):
        eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        eng.set_voice_values("frequency", plans["frequency"])
        eng.set_voice_taps(sample)
        for v in range(n):
            eng.schedule_voice_event("gate", v, int(plans["on_frame"][v]), float(plans["gate"][v]))
        silent = None
        buses, taps = [], []
        for f0 in range(0, total, blk):
            buses.append(eng.process_block(blk))
            taps.append(eng.read_voice_taps(blk))
        return np.concatenate(buses, axis=0), np.concatenate(taps, axis=1)

    bus_a, taps_a = run(block)
    bus_b, taps_b = run(512)
    # determinism + block-size independence at full size, bit-exact
    assert np.array_equal(taps_a, taps_b)
    assert np.array_equal(bus_a, bus_b)
    # voices are independent: a sampled voice of the big bank == the same voice in the oracle
    bank = ol.Bank(ol.BANK_FM, len(sample), SR)
    for i, v in enumerate(sample):
        bank.set_voice_frequency(i, float(plans["frequency"][v]))
    worst = 0.0
    for b, f0 in enumerate(range(0, total, block)):
        for i, v in enumerate(sample):
            on = int(plans["on_frame"][v])
            if f0 <= on < f0 + block:
                bank.push_event(i, on - f0, ol.EV_GATE, float(plans["gate"][v]))
        _, ref = bank.process_block(block, taps=list(range(len(sample))))
        worst = max(worst, rel_err(taps_a[:, f0:f0 + block], ref))
    observed.note(worst)
    assert worst <= TOL, worst
    # nothing sounds before the first note-on, and the bus is alive afterwards
    first_on = int(plans["on_frame"].min())
    assert np.all(bus_a[:first_on] == 0.0)
    assert np.max(np.abs(bus_a[1024:])) > 1.0


def test_zero_frame_block_is_a_noop_that_discards_its_events():
    """process_block(0) (the generated loop runs no frame; the block's event queues are cleared with it):
    nothing is rendered, a try_push'ed event is never delivered, later blocks are unaffected."""
    eng = oscen_amd.Engine("fm_voice", 3, sample_rate=SR)
    eng.set_voice_values("frequency", np.array([220.0, 330.0, 440.0], dtype=np.float32))
    assert eng.push_voice_event("gate", 1, 0, 1.0) == 0
    out = eng.process_block(0)
    assert out.shape == (0, 1) and eng.events_dropped == 1 and eng.frames_processed == 0
    assert np.all(eng.process_block(256) == 0.0)  # the note-on was discarded: the bank stays silent
    assert eng.push_voice_event("gate", 1, 3, 1.0) == 0
    assert np.abs(eng.process_block(256)).max() > 1e-3


def test_full_size_voice_shards_sum_to_the_whole_bank():
    """Row (e): voices are independent and note streams are keyed by global voice id, so the bus of the
    65 536-voice bank is the sum of the buses of its shards (here 4 ragged shards, as 4 ranks would hold
    them), up to the re-association of the bus sum; and a shard's voices are bit-identical to the same
    voices inside the whole bank."""
    n, total, block = 65536, 1024, 256
    cuts = [0, 16384, 32768 + 17, 49152 + 5, n]
    probe = np.array([3, 16383, 16384, 40000, n - 1], dtype=np.uint32)

    def run(lo, hi, taps):
        eng = oscen_amd.Engine("fm_voice", hi - lo, sample_rate=SR)
        plans = oscen_amd.note_plans(hi - lo, first_voice=lo)
        oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
        if len(taps):
            eng.set_voice_taps(taps)
        bus, tp = [], []
        for _ in range(total // block):
            bus.append(eng.process_block(block)[:, 0].astype(np.float64))
            if len(taps):
                tp.append(eng.read_voice_taps(block))
        return np.concatenate(bus), (np.concatenate(tp, axis=1) if len(taps) else None)

    whole, whole_taps = run(0, n, probe)
    parts = np.zeros_like(whole)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        local = np.array([v - lo for v in probe if lo <= v < hi], dtype=np.uint32)
        bus, tp = run(lo, hi, local)
        parts += bus
        rows = [i for i, v in enumerate(probe) if lo <= v < hi]
        if rows:
            assert np.array_equal(tp, whole_taps[rows])
    scale = max(1.0, float(np.max(np.abs(whole))))
    assert np.max(np.abs(parts - whole)) <= 1e-4 * scale  # f32 partial sums of ~6e4 terms, different trees
    assert np.max(np.abs(whole)) > 1.0


def test_render_equals_block_by_block_processing():
    """BlockRender::render(&[], tail) (oscen-lib/src/graph/offline.rs:46-90: a driver over process_block, "output
    is identical to realtime processing") == og_render: chunks of 512 frames with a ragged last chunk, bit for bit
    the same bus as calling process_block, for a mono and for the stereo (post-mix Tremolo) bank."""
    for graph, n in (("fm_voice", 96), ("epiano_voice", 12)):
        total = 512 * 3 + 77
        outs = []
        for mode in ("render", "blocks"):
            eng = oscen_amd.Engine(graph, n, sample_rate=SR)
            plans = oscen_amd.note_plans(n)
            oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
            if mode == "render":
                outs.append(eng.render(total, block=512))
            else:
                parts, left = [], total
                while left:
                    k = min(512, left)
                    parts.append(eng.process_block(k).copy())
                    left -= k
                outs.append(np.concatenate(parts, axis=0))
        assert outs[0].shape == (total, 2 if graph == "epiano_voice" else 1)
        assert np.array_equal(outs[0], outs[1])
        assert np.abs(outs[0]).max() > 1e-3
