"""Committed waveform fixtures (tests/golden/oracle_waveforms.npz, generator next to it).

CPU: the oracle reproduces them bit for bit (guards the oracle against drift).
GPU: the HIP kernels, through the C ABI, reproduce them within 1e-5 * max(1, |ref|) without running
the oracle at all.
"""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_oracle_waveforms", os.path.join(HERE, "golden", "make_oracle_waveforms.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
GOLD = np.load(os.path.join(HERE, "golden", "oracle_waveforms.npz"))


@pytest.mark.parametrize("name", sorted(gen.CASES))
def test_oracle_reproduces_committed_waveforms(name):
    got = gen.render_oracle(name)
    assert got.shape == GOLD[name].shape
    assert np.array_equal(got, GOLD[name]), "oracle output drifted from tests/golden/oracle_waveforms.npz"
    assert np.abs(got).max() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gen.CASES))
def test_kernels_match_committed_waveforms(name):
    import oscen_amd
    from tests import oracle_lib as ol

    kind, graph, n, blocks = gen.CASES[name]
    eng = oscen_amd.Engine(graph, n, sample_rate=gen.SR)
    eng.set_voice_values("frequency", np.asarray(gen.freqs(n), dtype=np.float32))
    eng.set_voice_taps(list(range(n)))
    if kind != ol.BANK_SAT4X:
        for fr, v, val in gen.score(n, blocks * 256):
            eng.schedule_voice_event("gate", v, fr, val)
    out = []
    for _ in range(blocks):
        eng.process_block(256)
        out.append(eng.read_voice_taps(256))
    got = np.concatenate(out, axis=1)
    ref = GOLD[name]
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    assert err <= 1e-5, err
