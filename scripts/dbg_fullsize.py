"""debug: error timeline of the worst sampled voices of the folded-plan fm bank (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oscen_amd
from tests import oracle_lib as ol
from tests.test_fullsize_gpu import oracle_taps, sample_voices

SR = 48000.0
total = 1024
n_big = 262144
taps = sample_voices(n_big)
vid = int(taps[94])
print("voice", vid)
for split in ("0", "2"):
    os.environ["OSCEN_GPU_SPLIT"] = split
    lo = max(0, vid - 20)
    n = 4096
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    plans = oscen_amd.note_plans(n, first_voice=lo, span=total)
    oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
    eng.set_voice_taps(np.arange(n, dtype=np.uint32))
    got = []
    for _ in range(total // 256):
        eng.process_block(256)
        got.append(eng.read_voice_taps(256))
    got = np.concatenate(got, axis=1)
    ref = oracle_taps(ol.BANK_FM, np.arange(lo, lo + n, dtype=np.uint32), total, 256)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    worst = np.argsort(err.max(axis=1))[::-1][:6]
    print("split", split, "depth", eng.pipeline_depth, "overall max", err.max(), "voices > 5e-6:", int((err.max(axis=1) > 5e-6).sum()),
          "> 1e-6:", int((err.max(axis=1) > 1e-6).sum()))
    for w in worst:
        e = err[w]
        first = int(np.argmax(e > 1e-6)) if (e > 1e-6).any() else -1
        print(" voice %d: max %.3g at %d, first>1e-6 at %d; plan on %d off %d retrig %d vel %.3f freq %.2f" % (
            lo + w, e.max(), int(e.argmax()), first, plans["on_frame"][w], plans["off_frame"][w], plans["retrig_frame"][w],
            plans["gate"][w], plans["frequency"][w]))
        k = int(e.argmax())
        for f in range(max(0, first - 2), min(total, first + 6)):
            print("    f %4d got % .8f ref % .8f err %.3g" % (f, got[w, f], ref[w, f], e[f]))
        print("    ... at max: got % .8f ref % .8f ; err profile every 64:" % (got[w, k], ref[w, k]),
              " ".join("%.1e" % e[i:i + 64].max() for i in range(0, total, 64)))
