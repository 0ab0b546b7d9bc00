import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oscen_amd
from tests import oracle_lib as ol
from tests.test_parity_gpu import Pair, midi_freqs, random_note_script, rel_err

def run(name, setup, nblocks=6, mid=None):
    n = 64
    p = Pair("fm_voice", ol.BANK_FM, n, ol.FM_PARAMS)
    setup(p)
    p.set_freqs(midi_freqs(n, 5))
    events = random_note_script(n, 4096, 6)
    f0 = 0
    errs = []
    for b in range(nblocks):
        if mid: mid(p, b)
        for fr, v, val in events:
            if f0 <= fr < f0 + 256:
                p.gate(v, fr - f0, val)
        bus, taps, ref_bus, ref_taps, ref64 = p.block(256)
        e = np.abs(taps - ref_taps) / np.maximum(1.0, np.abs(ref_taps))
        errs.append(float(e.max()))
        f0 += 256
    print(name, ["%.2e" % e for e in errs], flush=True)

run("baseline", lambda p: None)
run("fb3", lambda p: p.set_value_immediate("op3_feedback", 0.3))
run("fb2", lambda p: p.set_value_immediate("op2_feedback", 0.2))
run("route", lambda p: p.set_value_immediate("route", 0.5))
run("envamt", lambda p: p.set_value_immediate("filter_env_amount", 2000.0))
run("level", lambda p: p.set_value_immediate("op3_level", 0.8))
run("cutoff_imm", lambda p: p.set_value_immediate("filter_cutoff", 5000.0))
run("reso_imm", lambda p: p.set_value_immediate("filter_resonance", 2.5))
run("ramp_cutoff", lambda p: None, mid=lambda p, b: p.set_value("filter_cutoff", 6000.0) if b == 2 else None)
run("ramp_route", lambda p: None, mid=lambda p, b: p.set_value_with_ramp("route", 0.4, 300) if b == 2 else None)
run("ramp_level", lambda p: None, mid=lambda p, b: p.set_value("op2_level", 1.0) if b == 2 else None)
run("ramp_reso", lambda p: None, mid=lambda p, b: p.set_value("filter_resonance", 2.5) if b == 2 else None)
def four(p):
    p.set_value_immediate("op3_feedback", 0.3); p.set_value_immediate("op2_feedback", 0.2)
    p.set_value_immediate("route", 0.5); p.set_value_immediate("filter_env_amount", 2000.0)
run("four", four, nblocks=16)
def mid1(p, b):
    if b == 3:
        p.set_value("filter_cutoff", 6000.0); p.set_value_with_ramp("route", 0.1, 300)
run("four+ramps3", four, nblocks=16, mid=mid1)
def mid2(p, b):
    mid1(p, b)
    if b == 9:
        p.set_value_immediate("op3_level", 0.8); p.set_value("filter_resonance", 2.5)
run("four+ramps3+9", four, nblocks=16, mid=mid2)
run("two ramps", lambda p: None, nblocks=16, mid=mid1)
def mid3(p, b):
    if b == 3: p.set_value_with_ramp("route", 0.1, 300)
run("route ramp from .5", lambda p: p.set_value_immediate("route", 0.5), nblocks=8, mid=mid3)
