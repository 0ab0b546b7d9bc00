"""Play a short chord progression into a bank of FM voices through the MIDI front end and write a WAV.

    python examples/render_chords.py [out.wav]

raw MIDI bytes -> og_midi (parser, LRU voice allocator, note->Hz) -> per-voice frequency/gate events ->
fused voice kernel on the GPU -> mix bus -> 16-bit PCM.  Needs an MI355X (there is no CPU fallback).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oscen_amd  # noqa: E402

SR, BLOCK = 48000, 256
CHORDS = [(57, 60, 64, 69), (53, 57, 60, 65), (48, 52, 55, 60), (55, 59, 62, 67)]  # Am F C G


def main(path):
    eng = oscen_amd.Engine("fm_voice", 64, sample_rate=float(SR))
    eng.set_value_immediate("filter_cutoff", 2400.0)
    eng.set_value_immediate("filter_env_amount", 3000.0)
    midi = oscen_amd.Midi(eng, 64)
    out = []
    bar = SR  # one chord per second
    total_blocks = (len(CHORDS) * bar + SR) // BLOCK
    for b in range(total_blocks):
        f0 = b * BLOCK
        for ci, chord in enumerate(CHORDS):
            on, off = ci * bar, ci * bar + int(0.8 * bar)
            for k, note in enumerate(chord):
                t_on = on + k * 1200  # strum
                if f0 <= t_on < f0 + BLOCK:
                    midi.note_on(note, 70 + 10 * k, frame_offset=t_on - f0)
                if f0 <= off < f0 + BLOCK:
                    midi.note_off(note, frame_offset=off - f0)
        out.append(midi.process_block(BLOCK).copy())
    audio = np.concatenate(out, axis=0)
    peak = float(np.max(np.abs(audio)))
    audio = audio * np.float32(0.8 / max(peak, 1e-6))
    oscen_amd.write_wav(path, audio, sample_rate=SR, bits=16)
    print("wrote %s: %d frames, peak before normalisation %.3f" % (path, audio.shape[0], peak))
    return peak


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "chords.wav")
