"""og_group_voices (round 4): the voice SLOTS of a bank are re-ordered so that voices whose notes end together share waves;
voice NUMBERS do not change anywhere on the C ABI.  Every per-voice sample must be bit for bit the ungrouped bank's, the bus
may differ by the association of the sum only.  (`-m gpu`; also run on the host simulator by tests/test_hostsim_cpu.py.)"""
import numpy as np
import pytest

import oscen_amd

pytestmark = pytest.mark.gpu
SR = 48000.0


def fm_pair(n, total, group_policy=1, fold="slice"):
    plans = oscen_amd.note_plans(n, span=total, fold=fold)
    engs = []
    for grouped in (False, True):
        e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        oscen_amd.schedule_note_plans(e, plans, total_frames=total)
        if grouped:
            e.group_voices(group_policy)
        engs.append(e)
    return plans, engs[0], engs[1]


def test_grouped_bank_gives_the_same_voices_and_bus_as_the_ungrouped_bank():
    from tests import oracle_lib as ol

    n, block, blocks = 333, 256, 6
    total = block * blocks
    plans, plain, grouped = fm_pair(n, total)
    # the order is a permutation, and it follows the first note-off of each voice
    slots = np.array([grouped.voice_slot(v) for v in range(n)])
    assert sorted(slots) == list(range(n)) and not np.array_equal(slots, np.arange(n))
    ev_v, ev_f, ev_x = plans["events"]
    first_off = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    off = (ev_x <= 0) & (ev_f < total)
    np.minimum.at(first_off, ev_v[off], ev_f[off])
    assert np.all(np.diff(first_off[np.argsort(slots)]) >= 0)
    assert [plain.voice_slot(v) for v in (0, 5, n - 1)] == [0, 5, n - 1]
    taps = np.arange(0, n, 5, dtype=np.uint32)
    for e in (plain, grouped):
        e.set_voice_taps(taps)
    rng = np.random.default_rng(5)
    bank = ol.Bank(ol.BANK_FM, n, SR)
    for v in range(n):
        bank.set_voice_frequency(v, float(plans["frequency"][v]))
    worst = 0.0
    for b in range(blocks):
        if b == 2:  # live traffic on top of the score, by VOICE NUMBER: retriggers, and new frequencies for a range of voices
            for v in rng.choice(n, 40, replace=False):
                for e in (plain, grouped):
                    e.push_voice_event("gate", int(v), int(v) % block, 0.7)
            f_new = (220.0 + 3.0 * np.arange(150)).astype(np.float32)
            for e in (plain, grouped):
                e.set_voice_values("frequency", f_new, first=100)       # a range: scattered over the slots
                e.set_voice_values("frequency", f_new[:3] * 2.0, first=5)  # a few values: word by word
        bus_a = plain.process_block(block)
        bus_b = grouped.process_block(block)
        ta, tb = plain.read_voice_taps(block), grouped.read_voice_taps(block)
        assert np.array_equal(ta, tb), b  # per-voice samples: bit for bit
        scale = max(1.0, float(np.abs(ta).sum(axis=0).max()) * n / len(taps))
        assert np.max(np.abs(bus_a - bus_b)) <= 4e-6 * scale, b  # the bus: association of the sum only
        worst = max(worst, float(np.abs(bus_a).max()))
    assert worst > 1.0
    # node state by voice number
    for path in ("op1_osc.phase", "env1.level"):
        a = plain.read_state_field(path)
        assert np.any(a != 0.0)
        assert np.array_equal(a, grouped.read_state_field(path)), path
        assert np.array_equal(a[40:90], grouped.read_state_field(path, first=40, n=50)), path


def test_policy_two_deals_the_waves_out_and_changes_no_voice():
    """policy 2 = policy 1 + the groups of 64 slots placed so that groups i, i + 256, ... carry about the same number of
    events (banks of >= 32 768 voices); still a permutation, still the same voices"""
    n, block, blocks = 33000, 128, 2
    total = block * blocks
    plans = oscen_amd.note_plans(n, span=total, fold="slice")
    engs = []
    for policy in (0, 1, 2):
        e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        oscen_amd.schedule_note_plans(e, plans, total_frames=total)
        e.group_voices(policy)
        engs.append(e)
    probe = np.unique(np.linspace(0, n - 1, 200).astype(np.uint32))
    slots = [np.array([e.voice_slot(int(v)) for v in range(0, n, 7)]) for e in engs]
    assert np.array_equal(slots[0], np.arange(0, n, 7)) and not np.array_equal(slots[1], slots[2])
    all2 = np.array([engs[2].voice_slot(v) for v in range(n)])
    assert np.array_equal(np.sort(all2), np.arange(n))
    for e in engs:
        e.set_voice_taps(probe)
    for _ in range(blocks):
        buses = [e.process_block(block) for e in engs]
        taps = [e.read_voice_taps(block) for e in engs]
        assert np.array_equal(taps[0], taps[1]) and np.array_equal(taps[0], taps[2])
        scale = max(1.0, float(np.abs(buses[0]).max()))
        assert np.max(np.abs(buses[0] - buses[1])) <= 1e-4 * scale and np.max(np.abs(buses[0] - buses[2])) <= 1e-4 * scale


def test_grouped_bank_against_the_oracle():
    from tests import oracle_lib as ol
    from tests.test_fullsize_gpu import oracle_taps

    n, block, blocks = 200, 256, 4
    total = block * blocks
    plans, _, grouped = fm_pair(n, total, fold="scale")  # (the fold oracle_taps replays)
    voices = np.arange(3, n, 11, dtype=np.uint32)
    grouped.set_voice_taps(voices)
    got = []
    for _ in range(blocks):
        grouped.process_block(block)
        got.append(grouped.read_voice_taps(block))
    got = np.concatenate(got, axis=1)
    ref = oracle_taps(ol.BANK_FM, voices, total, block)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert float(err.max()) <= 1e-5, float(err.max())


def test_grouping_is_refused_once_the_bank_has_rendered_and_policy_zero_restores_the_identity():
    n = 130
    plans, plain, grouped = fm_pair(n, 1024)
    grouped.group_voices(0)
    assert [grouped.voice_slot(v) for v in range(n)] == list(range(n))
    grouped.group_voices(1)
    grouped.group_voices(1)  # (idempotent: the key is computed on voice numbers)
    a, b = plain.process_block(256), grouped.process_block(256)
    assert np.allclose(a, b, atol=1e-4)
    with pytest.raises(oscen_amd.OscenError) as ei:
        grouped.group_voices(1)
    assert ei.value.args[0].startswith("oscen_gpu error %d" % oscen_amd.OG_E_STATE)
    with pytest.raises(oscen_amd.OscenError):
        plain.group_voices(7)
    e = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    e.set_voice_taps([1, 2, 3])
    with pytest.raises(oscen_amd.OscenError):
        e.group_voices(1)


def test_a_snapshot_of_a_grouped_bank_carries_the_voice_order():
    n, block = 150, 128
    total = block * 8
    plans, plain, grouped = fm_pair(n, total)
    taps = np.arange(0, n, 7, dtype=np.uint32)
    grouped.set_voice_taps(taps)
    for _ in range(3):
        grouped.process_block(block)
    blob = grouped.save_state()
    cont = []
    for _ in range(5):
        cont.append((grouped.process_block(block).copy(), grouped.read_voice_taps(block)))
    fresh = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    fresh.load_state(blob)
    assert [fresh.voice_slot(v) for v in range(n)] == [grouped.voice_slot(v) for v in range(n)]
    fresh.set_voice_taps(taps)
    for bus, tp in cont:
        b2 = fresh.process_block(block)
        assert np.array_equal(b2, bus) and np.array_equal(fresh.read_voice_taps(block), tp)
    # a version-2 blob (ungrouped engine) loaded into the grouped engine restores the identity
    blob2 = plain.save_state()
    grouped.load_state(blob2)
    assert [grouped.voice_slot(v) for v in (0, 7, n - 1)] == [0, 7, n - 1]
    bad = blob.copy()
    bad[-4:] = 255  # the last entry of the order out of range
    with pytest.raises(oscen_amd.OscenError):
        fresh.load_state(bad)


def test_voice_taps_survive_a_load_that_changes_the_slot_order():
    """og_load_state adopts the blob's voice order; taps set BEFORE the load keep reading the same voice numbers (ADVICE r4:
    round 4 silently cleared them when the order changed)."""
    n, block = 150, 128
    total = block * 8
    plans, plain, grouped = fm_pair(n, total)
    taps = np.arange(3, n, 11, dtype=np.uint32)
    for _ in range(3):
        grouped.process_block(block)
    blob = grouped.save_state()
    grouped.set_voice_taps(taps)
    want = [(grouped.process_block(block).copy(), grouped.read_voice_taps(block)) for _ in range(4)]
    fresh = oscen_amd.Engine("fm_voice", n, sample_rate=SR)   # identity order
    fresh.set_voice_taps(taps)                                # ... taps resolved under the identity
    fresh.load_state(blob)                                    # the order changes under them
    assert fresh.voice_slot(int(taps[1])) == grouped.voice_slot(int(taps[1]))
    for bus, tp in want:
        b2 = fresh.process_block(block)
        assert np.array_equal(b2, bus)
        assert np.array_equal(fresh.read_voice_taps(block), tp)
    assert np.abs(want[0][1]).max() > 0.0


def test_event_outputs_and_cluster_shards_keep_voice_numbers_under_grouping():
    oscen_amd.register_node(
        "GvBurst::new", inputs=[("period", "value", 100.0, 0), ("trig", "event", 0.0, -1)], outputs=["level"], n_ctor_args=1,
        state=[("count", "u32", 0, -1), ("fired", "f32", 0.0, -1)], event_outputs=["tick"],
        handlers={"trig": "    count = 0u;\n"},
        process="""
    count += 1u;
    if ((float)count >= period) {
        count = 0u;
        fired += 1.0f;
        tick.push(fired);
    }
    level = fired;
""")
    try:
        g = oscen_amd.Graph("gv_bursts")
        g.input_value("period", 100.0, per_voice=True)
        g.input_event("trig")
        g.output_stream("out")
        g.output_event("ticks")
        g.node("b", "GvBurst::new", 100.0)
        g.connect("period", "b.period")
        g.connect("trig", "b.trig")
        g.connect("b.level", "out")
        g.connect("b.tick", "ticks")
        n = 200
        periods = (37 + (np.arange(n) * 7) % 90).astype(np.float32)
        rng = np.random.default_rng(9)
        ev_v = rng.integers(0, n, 300).astype(np.uint32)
        ev_f = rng.integers(0, 700, 300).astype(np.uint64)
        ev_x = np.where(rng.random(300) < 0.5, 0.0, 1.0).astype(np.float32)  # half of them "note-offs" (value <= 0): the grouping key
        made = []
        for kind in ("plain", "grouped", "cluster"):
            x = oscen_amd.Cluster(g, n, [0, 0, 0], sample_rate=SR) if kind == "cluster" else oscen_amd.Engine(g, n, sample_rate=SR)
            x.set_voice_values("period", periods)
            x.schedule_voice_events("trig", ev_v, ev_f, ev_x)
            if kind != "plain":
                x.group_voices(1)
            made.append(x)
        plain, grouped, cl = made
        assert sorted(grouped.voice_slot(v) for v in range(n)) == list(range(n))
        assert any(grouped.voice_slot(v) != v for v in range(n))
        for frames in (256, 100, 412):
            a, b, c = plain.process_block(frames), grouped.process_block(frames), cl.process_block(frames)
            assert np.allclose(a, b, rtol=1e-6, atol=1e-5) and np.allclose(a, c, rtol=1e-6, atol=1e-5)
        ev1, _ = plain.read_output_events()
        ev2, _ = grouped.read_output_events()
        ev3, _ = cl.read_output_events()
        assert len(ev1) > 500 and np.array_equal(ev1, ev2) and np.array_equal(ev1, ev3)
        assert np.array_equal(plain.read_state_field("b.fired"), grouped.read_state_field("b.fired"))
    finally:
        oscen_amd.unregister_node("GvBurst::new")
