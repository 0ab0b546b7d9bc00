"""Parity on EXACTLY the configuration bench.py times (BASELINE config 2), for the whole horizon.

The timed path: 65 536 fm-synth voices, voice slots grouped by first note-off (`og_group_voices(1)`, bench.py's default), the
kernel shape the engine picks at that size (the wide four-wave pipeline `og_k4w_*`), the blocks handed over by
`og_process_blocks_async` in 20-block runs with a flush after each (a timed region of the driver's command), the score
resident in HBM.  Three scores:
  * SURVEY 8(d) config 2 as written: 48 000 frames = 187 blocks of 256 + one of 128; note-on in block 0, note-off in
    [12 000, 36 000], second note-on in [36 001, 44 000] -- AdsrEnvelope's retrigger from Release
    (oscen-lib/src/envelope/adsr.rs:250-273);
  * the cyclic fold the 20-step command plays (every voice's 1 s plan from a per-voice offset), over the 109 blocks of a
    `--steps 20 --warmup 5 --repeats 5` run;
  * config 2's VARIANT: op3_feedback .3, op2_feedback .2, route .5, filter_env_amount 2000 (per-sample tan,
    oscen-lib/src/filters/tpt/mod.rs:85-101) and `set_filter_cutoff(6000)` -- `[ramp: 2205]`,
    examples/fm-synth/src/fm_voice.rs:10-48 -- right before frame 4 800 (the block is cut there: 18 x 256 + 192).

Per-voice samples cannot be read out of a queued launch (taps make the engine launch per block), so each score runs twice
on the SAME grouped bank and kernel shape:
  (checked) blocking `og_process_block` per block with >= 128 tapped voices  -> every tapped sample within
            1e-5 * max(1, |ref|) of the oracle fed the identical events; the error-vs-time curve is recorded;
  (timed)   the path above                                                    -> its bus equals the checked run's bus BIT
            FOR BIT in every frame and its complete DSP state after the run equals the checked run's bit for bit (same
            bus + same final state of every voice = same trajectory), and the bus is within the re-association bound of
            the oracle's f64 sum over all 65 536 voices.
"""
import json
import os

import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol
from tests.test_fullsize_gpu import DeviceBuffer, dsp_state

pytestmark = pytest.mark.gpu

SR = 48000.0
TOL = 1e-5
N = 65536
RUN = 20  # blocks per og_process_blocks_async call / flush: a timed region of the driver's command

SURVEY2 = {"op3_feedback": 0.3, "op2_feedback": 0.2, "route": 0.5, "filter_env_amount": 2000.0}


def sampled_voices(n, k=136):
    rng = np.random.default_rng(0x06CE)
    fixed = [0, 1, 63, 64, 65, n // 2 - 1, n // 2, n - 65, n - 64, n - 1]
    return np.unique(np.concatenate([rng.integers(0, n, k), fixed])).astype(np.uint32)


def block_list(total, cut=None):
    """256-frame blocks covering `total` frames; a boundary at frame `cut` (setters act between blocks)."""
    out, f = [], 0
    while f < total:
        b = min(256, total - f)
        if cut is not None and f < cut < f + b:
            b = cut - f
        out.append(b)
        f += b
    return out


def make_engine(plans, total, sets):
    eng = oscen_amd.Engine("fm_voice", N, sample_rate=SR)
    for k, v in sets.items():
        eng.set_value_immediate(k, v)
    oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
    eng.group_voices(1)
    return eng


def checked_run(plans, total, blocks, sets, ramp, taps):
    eng = make_engine(plans, total, sets)
    eng.set_voice_taps(taps)
    bus, tp, f = [], [], 0
    for b in blocks:
        if ramp and f == ramp[2]:
            eng.set_value(ramp[0], ramp[1])
        bus.append(eng.process_block(b).copy())
        tp.append(eng.read_voice_taps(b))
        f += b
    info = {"variant": eng.kernel_variant, "depth": eng.pipeline_depth}
    state = dsp_state(eng)
    eng.close()
    return np.concatenate(bus, axis=0), np.concatenate(tp, axis=1), state, info


def timed_run(plans, total, blocks, sets, ramp):
    eng = make_engine(plans, total, sets)
    ch = eng.channels
    dev = DeviceBuffer(total * ch * 4)
    eng.set_bus_batching(0)  # queued blocks share a launch, the engine picks how many (bench.py's default)
    eng.enable_kernel_timing(True)
    i, f, calls = 0, 0, 0
    while i < len(blocks):
        if ramp and f == ramp[2]:
            eng.set_value(ramp[0], ramp[1])
        j = i
        while j < len(blocks) and j - i < RUN and blocks[j] == blocks[i] and not (ramp and j > i and f + (j - i) * blocks[i] == ramp[2]):
            j += 1
        eng.process_blocks_async(blocks[i], j - i, dev.ptr.value + f * ch * 4, blocks[i] * ch * 4)
        calls += 1
        f += (j - i) * blocks[i]
        i = j
        eng.flush()
    eng.synchronize()
    _, launches = eng.kernel_time_ms()
    out = dev.to_host().reshape(total, ch)
    dev.free()
    info = {"variant": eng.kernel_variant, "depth": eng.pipeline_depth, "launches": launches, "calls": calls}
    state = dsp_state(eng)
    eng.close()
    return out, state, info


def oracle_taps(voices, plans, total, blocks, sets, ramp):
    """the sampled voices in a small oracle bank fed the SAME events (taken from the arrays the engine was given)"""
    bank = ol.Bank(ol.BANK_FM, len(voices), SR)
    for k, v in sets.items():
        bank.set_value_immediate(ol.FM_PARAMS.index(k), v)
    ev_v, ev_f, ev_x = plans["events"]
    per_voice = []
    for i, v in enumerate(voices):
        bank.set_voice_frequency(i, float(plans["frequency"][v]))
        m = ev_v == v
        per_voice.append(sorted(zip(ev_f[m].tolist(), range(int(m.sum())), ev_x[m].tolist())))
    out, f = [], 0
    for b in blocks:
        if ramp and f == ramp[2]:
            bank.set_value(ol.FM_PARAMS.index(ramp[0]), ramp[1])
        for i, evs in enumerate(per_voice):
            for fr, _, val in evs:
                if f <= fr < f + b:
                    bank.push_event(i, fr - f, ol.EV_GATE, float(val))
        _, t = bank.process_block(b, taps=list(range(len(voices))))
        out.append(t)
        f += b
    return np.concatenate(out, axis=1)


SCORES = {
    # name: (frames, span, fold, parameter set, ramp)
    "config2_one_second": (48000, 48000, "slice", {}, None),
    "config2_cyclic_fold_109_blocks": (109 * 256, 109 * 256, "cyclic", {}, None),
    "config2_variant_one_second": (48000, 48000, "slice", SURVEY2, ("filter_cutoff", 6000.0, 4800)),
}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("name", list(SCORES))
def test_the_timed_configuration_against_the_oracle_over_its_whole_horizon(name):
    total, span, fold, sets, ramp = SCORES[name]
    plans = oscen_amd.note_plans(N, span=span, fold=fold)
    blocks = block_list(total, ramp[2] if ramp else None)
    assert sum(blocks) == total
    taps = sampled_voices(N)
    assert len(taps) >= 128

    # (checked) grouped bank, the engine's kernel shape, blocking with taps: sampled voices against the oracle
    bus_c, tp, state_c, info_c = checked_run(plans, total, blocks, sets, ramp, taps)
    assert info_c["depth"] == 4 and info_c["variant"].startswith("og_k4w_"), info_c
    ref = oracle_taps(taps, plans, total, blocks, sets, ramp)
    err = np.abs(tp - ref) / np.maximum(1.0, np.abs(ref))
    worst = float(err.max())
    observed.note(worst)
    assert np.abs(ref).max() > 0.05 and np.isfinite(tp).all()
    # error-vs-time curve: max over the sampled voices per block
    edges = np.cumsum([0] + blocks)
    curve = [(int(edges[i]), float(err[:, edges[i]:edges[i + 1]].max())) for i in range(len(blocks))]
    path = os.environ.get("OSCEN_OBSERVED")
    if path:
        with open("%s.timed_%s.json" % (path, name), "w") as f:
            json.dump({"voices": N, "sampled": int(len(taps)), "frames": total, "curve": curve, "worst": worst,
                       "ref_peak": float(np.abs(ref).max()), "kernel": info_c["variant"]}, f)
    assert worst <= TOL, (worst, max(curve, key=lambda c: c[1]))
    if total == 48000 and not sets:
        # the plan really exercised retrigger-from-release on the sampled voices: some voice's second note-on falls where its
        # release (0.3 s default, fm_voice.rs) has not finished
        off, re = plans["off_frame"][taps], plans["retrig_frame"][taps]
        assert np.any(re - off < 0.3 * SR)

    # (timed) og_process_blocks_async in 20-block runs: same bus, same final state, bit for bit
    bus_t, state_t, info_t = timed_run(plans, total, blocks, sets, ramp)
    assert info_t["depth"] == 4 and info_t["variant"].startswith("og_k4w_"), info_t
    assert info_t["launches"] < len(blocks) // 4, info_t  # queued: many blocks per launch
    assert np.array_equal(bus_t, bus_c), (info_t, float(np.abs(bus_t - bus_c).max()))
    assert np.array_equal(state_t, state_c), info_t
    # the bus against the oracle's f64 sum over every voice (the reference's sequential f32 fold, emit_node.rs:463-466, and
    # the fixed tree differ by re-association only)
    threads = min(os.cpu_count() or 1, 32)
    mono, abs_sum, secs = ol.render_mt(ol.BANK_FM, 0, N, total, block=256, threads=threads, group=8, seed=oscen_amd.SYNTH_SEED,
                                       span=span if span < 48000 else 0, fold="slice" if fold == "cyclic" else "scale",
                                       sets=[(ol.FM_PARAMS.index(k), v) for k, v in sets.items()],
                                       ramps=[(ol.FM_PARAMS.index(ramp[0]), ramp[1], ramp[2])] if ramp else [])
    diff = np.abs(bus_t[:, 0].astype(np.float64) - mono)
    assert np.all(diff <= 2e-6 * abs_sum + 1e-5), float((diff / (abs_sum + 1e-30)).max())
    assert np.abs(mono).max() > 1.0
    print("\n%s: %d sampled voices max rel err %.3g (curve peak at frame %d), bus max |err|/sum|x| %.3g, oracle %.1f s, %s, %d launches / %d blocks"
          % (name, len(taps), worst, max(curve, key=lambda c: c[1])[0], float((diff / (abs_sum + 1e-30)).max()), secs, info_t["variant"],
             info_t["launches"], len(blocks)))
