"""Oversampled (`* N`) nodes and cross-rate edges on the GPU vs the oracle's resampler kernels
(oscen-lib/src/resample/*, generated multirate body emit_frame.rs:114-176)."""
import ctypes as C

import numpy as np
import pytest

import oscen_amd
from tests import observed
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
SR = 48000.0


def rel_err(got, ref):
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))


@pytest.mark.parametrize("graph,kind", [("sat4x_voice", ol.BANK_SAT4X), ("sat1x_voice", ol.BANK_SAT1X)])
def test_saturator_bank_parity(graph, kind):
    # BASELINE.json configs[4]: PolyBLEP saw *4 -> HardClip *4 -> [sinc] -> out, per-voice frequency
    n = 130
    rng = np.random.default_rng(21)
    freqs = np.exp(rng.uniform(np.log(100.0), np.log(4000.0), n)).astype(np.float32)
    eng = oscen_amd.Engine(graph, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    bank = ol.Bank(kind, n, SR)
    for v in range(n):
        bank.set_voice_frequency(v, float(freqs[v]))
    assert eng.latency_samples == (8 if kind == ol.BANK_SAT4X else 0)   # 33 / 4 (emit_struct.rs:534-570)
    worst = 0.0
    for frames in (256, 256, 512, 100, 412):
        bus = eng.process_block(frames)
        got = eng.read_voice_taps(frames)
        ref_bus, ref = bank.process_block(frames, taps=list(range(n)))
        worst = max(worst, rel_err(got, ref))
        scale = max(1.0, float(np.max(np.sum(np.abs(ref), axis=0))))
        assert np.max(np.abs(bus[:, 0] - bank.last_bus_f64(frames))) <= 1e-5 * scale
    assert np.max(np.abs(ref)) > 0.3
    observed.note(worst)
    assert worst <= 1e-5, worst


def _oracle_chain(lib, n_frames, freq, up_kind, down_kind, N):
    """sine(outer) -> [up] -> HardClip*N -> [down] -> out, composed from the oracle's kernels"""
    osc = ol.PolyBlep()
    lib.oo_polyblep_new(C.byref(osc), freq, 0.9, ol.PB_SINE)
    osc.sample_rate = SR
    if up_kind == "sinc":
        up = ol.SincUp(); lib.oo_sinc_up_new(C.byref(up), N); upf = lib.oo_sinc_up_process
    elif up_kind == "sinc_iir":
        up = ol.IirResampler(); lib.oo_iir_resampler_new(C.byref(up), N); upf = lib.oo_iir_up_process
    else:
        up = ol.LinearUp(); lib.oo_linear_up_new(C.byref(up), N); upf = lib.oo_linear_up_process
    if down_kind == "sinc":
        dn = ol.SincDown(); lib.oo_sinc_down_new(C.byref(dn), N)
        downf = lambda xs: lib.oo_sinc_down_process(C.byref(dn), ol.fptr(xs))
    elif down_kind == "sinc_iir":
        dn = ol.IirResampler(); lib.oo_iir_resampler_new(C.byref(dn), N)
        downf = lambda xs: lib.oo_iir_down_process(C.byref(dn), ol.fptr(xs))
    elif down_kind == "linear":
        downf = lambda xs: lib.oo_linear_down_process(N, ol.fptr(xs))
    else:
        downf = lambda xs: lib.oo_latch_down_process(N, ol.fptr(xs))
    clip = lambda x: float(min(max(np.float32(x) * np.float32(1.5), np.float32(-0.7)), np.float32(0.7)))
    out = np.zeros(n_frames, dtype=np.float32)
    buf, dbuf = np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.float32)
    for i in range(n_frames):
        lib.oo_polyblep_process(C.byref(osc))
        upf(C.byref(up), osc.output, ol.fptr(buf))
        for j in range(N):
            dbuf[j] = clip(buf[j])
        out[i] = downf(dbuf)
    return out


@pytest.mark.parametrize("up_kind,down_kind,N", [("sinc", "sinc", 4), ("sinc_iir", "sinc_iir", 2),
                                                 ("linear", "linear", 8), ("sinc", "latch", 2)])
def test_cross_rate_edge_kernels_via_jit(up_kind, down_kind, N):
    # multirate_graph.rs ClipOversampled shape, every policy; compiled with hiprtc (not a built-in graph)
    lib = ol.load()
    g = oscen_amd.Graph("clip_os_%s_%s_%d" % (up_kind, down_kind, N))
    g.input_value("frequency", 440.0, per_voice=True)
    g.output_stream("out")
    g.node("src", "PolyBlepOscillator::sine", 440.0, 0.9)
    g.node("clip", "HardClip::new", rate=N)
    g.connect("frequency", "src.frequency")
    g.connect("src.output", "clip.input", up_kind)
    g.connect("clip.output", "out", down_kind)
    n, frames = 8, 256
    freqs = np.array([110.0, 440.0, 997.0, 2000.0, 4800.0, 9600.0, 13000.0, 55.0], dtype=np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    got = []
    for _ in range(3):
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
    got = np.concatenate(got, axis=1)
    worst = 0.0
    for v in range(n):
        ref = _oracle_chain(lib, 3 * frames, float(freqs[v]), up_kind, down_kind, N)
        worst = max(worst, rel_err(got[v], ref))
    observed.note(worst)
    assert worst <= 1e-5, worst


def test_delay_and_feedback_edge_inside_an_oversampled_region():
    """`Delay::new(..) * 2` ticks at the oversampled rate (ring sized from sr * 2, delay/mod.rs:59-69) and closes a
    feedback loop whose both ends are inner-rate nodes (ir/lower.rs:580-652 allows it): the consumer scheduled before
    the delay reads what the delay produced on the previous INNER tick."""
    from tests.graph_interp import VoiceInterp

    g = oscen_amd.Graph("os_echo")
    g.input_value("frequency", 220.0, per_voice=True)
    g.input_value("fb", 0.45)
    g.output_stream("out")
    g.node("osc", "PolyBlepOscillator::saw", 220.0, 0.3)
    g.node("mix", "Mixer::new", rate=2)
    g.node("clip", "HardClip::new", rate=2)
    g.node("fbk", "Gain::new", 0.4, rate=2)
    g.node("d", "Delay::new", 37.0, 0.2, rate=2)
    g.connect("frequency", "osc.frequency")
    g.connect("osc.output", "mix.input_a", "linear")
    g.connect("fb", "fbk.gain")
    g.connect("mix.output", "clip.input")
    g.connect_via("clip.output", "d", "fbk.input")
    g.connect("fbk.output", "mix.input_b")
    g.connect("clip.output", "out", "sinc")
    src = g.kernel_source()
    order = [ln for ln in src.splitlines() if ln.startswith("// Node order:")][0].split()[3:]
    assert order.index("fbk") < order.index("d")  # the consumer of the feedback edge runs first
    n, frames, blocks = 5, 160, 4
    freqs = np.array([98.0, 220.0, 587.33, 1318.5, 2093.0], dtype=np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    desc = {"inputs": [("frequency", "value", 220.0, 0), ("fb", "value", 0.45, 0)],
            "nodes": [("osc", "PolyBlepOscillator::saw", [220.0, 0.3]), ("mix", "Mixer::new", []), ("clip", "HardClip::new", []),
                      ("fbk", "Gain::new", [0.4]), ("d", "Delay::new", [37.0, 0.2])],
            "edges": [("frequency", "osc.frequency"), ("osc.output", "mix.input_a"), ("fb", "fbk.gain"),
                      ("mix.output", "clip.input"), ("clip.output", "d.input"), ("d.output", "fbk.input"),
                      ("fbk.output", "mix.input_b"), ("clip.output", "out")],
            "order": order, "rates": {"mix": 2, "clip": 2, "fbk": 2, "d": 2}, "policies": {1: "linear", 7: "sinc"}}
    voices = [VoiceInterp(desc, SR, {"frequency": float(freqs[v])}) for v in range(n)]
    got, ref = [], np.zeros((n, frames * blocks), dtype=np.float32)
    for b in range(blocks):
        if b == 2:
            eng.set_value("fb", 0.6)
            for vi in voices:
                vi.set_value("fb", 0.6)
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
        for v, vi in enumerate(voices):
            for i in range(frames):
                ref[v, b * frames + i] = vi.frame([])
    got = np.concatenate(got, axis=1)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-2
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    observed.note(err)
    assert err <= 1e-5, err
