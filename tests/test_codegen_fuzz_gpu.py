"""Random voice graphs end to end: graph compiler -> hiprtc -> MI355X, against a per-sample interpreter of
the same description over the oracle's nodes (tests/graph_interp.py).  Checks the COMPILER (schedule,
fan-in sums, compound sources, feedback edges, hoisting, ramp tables, pipelined variants) on graphs
nobody hand-expanded."""
import numpy as np
import pytest

import oscen_amd
from tests.graph_interp import VoiceInterp
from tests.test_codegen_fuzz_cpu import random_graph

pytestmark = pytest.mark.gpu
SR = 48000.0


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_random_graph_matches_the_node_interpreter(seed):
    g = random_graph(seed)
    desc = g.description()
    n, frames, blocks = 5, 200, 4
    freqs = np.array([82.41, 146.83, 220.0, 329.63, 523.25], dtype=np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    on = [3 + 11 * v for v in range(n)]
    off = [430 + 7 * v for v in range(n)]
    for v in range(n):
        eng.schedule_voice_event("gate", v, on[v], 0.9)
        eng.schedule_voice_event("gate", v, off[v], 0.0)
    voices = [VoiceInterp(desc, SR, {"frequency": float(freqs[v])}) for v in range(n)]
    got, ref = [], np.zeros((n, frames * blocks), dtype=np.float32)
    for b in range(blocks):
        if b == 1:
            eng.set_value("cutoff", 2600.0)
            for vi in voices:
                vi.set_value("cutoff", 2600.0)
        if b == 2:
            eng.set_value("amount", 0.2)
            for vi in voices:
                vi.set_value("amount", 0.2)
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
        for v, vi in enumerate(voices):
            for i in range(frames):
                f = b * frames + i
                gates = [("gate", 0.9)] if f == on[v] else ([("gate", 0.0)] if f == off[v] else [])
                ref[v, f] = vi.frame(gates)
    got = np.concatenate(got, axis=1)
    assert np.isfinite(ref).all()
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    assert err <= 1e-5, (seed, err, desc["order"])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_multirate_graph_matches_the_node_interpreter(seed):
    """Oversampled regions: `* N` nodes, Up/Down edges with every policy (and the default), value ports and
    explicit [latch] edges that stay un-resampled, outer nodes downstream of the region (row a4, a18-a20)."""
    from tests.test_codegen_fuzz_cpu import random_multirate_graph
    g = random_multirate_graph(seed)
    desc = g.description()
    n, frames, blocks = 4, 160, 3
    freqs = np.array([98.0, 220.0, 587.33, 1318.5], dtype=np.float32)
    eng = oscen_amd.Engine(g, n, sample_rate=SR)
    eng.set_voice_values("frequency", freqs)
    eng.set_voice_taps(list(range(n)))
    on = [2 + 9 * v for v in range(n)]
    off = [300 + 5 * v for v in range(n)]
    for v in range(n):
        eng.schedule_voice_event("gate", v, on[v], 0.8)
        eng.schedule_voice_event("gate", v, off[v], 0.0)
    voices = [VoiceInterp(desc, SR, {"frequency": float(freqs[v])}) for v in range(n)]
    got, ref = [], np.zeros((n, frames * blocks), dtype=np.float32)
    for b in range(blocks):
        if b == 1:
            eng.set_value("cutoff", 4200.0)
            for vi in voices:
                vi.set_value("cutoff", 4200.0)
        eng.process_block(frames)
        got.append(eng.read_voice_taps(frames))
        for v, vi in enumerate(voices):
            for i in range(frames):
                f = b * frames + i
                gates = [("gate", 0.8)] if f == on[v] else ([("gate", 0.0)] if f == off[v] else [])
                ref[v, f] = vi.frame(gates)
    got = np.concatenate(got, axis=1)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-3
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    assert err <= 1e-5, (seed, err, desc["order"], desc["rates"], desc["policies"])
