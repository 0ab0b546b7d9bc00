"""Regenerate tests/golden/oracle_waveforms.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_oracle_waveforms.py`).

These are OUTPUTS OF THE ORACLE, not of the reference (the reference is nightly Rust and cannot be
built in this image): they pin the oracle against accidental drift (CPU test, bit-exact) and give
the GPU suite a committed target that does not depend on the oracle being executed on the GPU box.
Per-voice waveforms of the four bank kinds on a fixed little score.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as ol  # noqa: E402

SR = 48000.0
# kind -> (graph name of the engine, voices, blocks of 256 frames)
CASES = {"fm": (ol.BANK_FM, "fm_voice", 4, 24), "sub": (ol.BANK_SUB, "sub_voice", 4, 16),
         "epiano": (ol.BANK_EPIANO, "epiano_voice", 2, 8), "sat4x": (ol.BANK_SAT4X, "sat4x_voice", 2, 4)}
NOTES = [45, 57, 69, 81]


def score(n_voices, total):
    """(frame, voice, gate value): note on at 10 + 37 v, off at 55% of the render, retrigger at 80%."""
    ev = []
    for v in range(n_voices):
        vel = (60 + 20 * v) / 127.0
        ev += [(10 + 37 * v, v, vel), (int(total * 0.55) + 11 * v, v, 0.0), (int(total * 0.8) + 5 * v, v, vel)]
    return sorted(ev)


def freqs(n_voices):
    return [float(np.float32(440.0) * np.float32(2.0) ** np.float32((NOTES[v % 4] - 69) / 12.0)) for v in range(n_voices)]


def render_oracle(name):
    kind, _, n, blocks = CASES[name]
    bank = ol.Bank(kind, n, SR)
    for v, f in enumerate(freqs(n)):
        bank.set_voice_frequency(v, f)
    ev = score(n, blocks * 256) if kind != ol.BANK_SAT4X else []
    out = []
    for b in range(blocks):
        for fr, v, val in ev:
            if b * 256 <= fr < (b + 1) * 256:
                bank.push_event(v, fr - b * 256, ol.EV_GATE, val)
        _, taps = bank.process_block(256, taps=list(range(n)))
        out.append(taps)
    return np.concatenate(out, axis=1)


if __name__ == "__main__":
    data = {name: render_oracle(name) for name in CASES}
    path = os.path.join(ROOT, "tests", "golden", "oracle_waveforms.npz")
    np.savez_compressed(path, **data)
    for k, v in data.items():
        print(k, v.shape, float(np.abs(v).max()))
    print("wrote", path, os.path.getsize(path), "bytes")
