"""Where does a realistic note stream cost kernel time?  fm_voice, 65536 voices, 25 blocks of 256 (5 warm-up)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oscen_amd

N, BLOCK, W, K = 65536, 256, 5, 20
rng = np.random.default_rng(1)
plans = oscen_amd.note_plans(N, span=0)
freq = plans["frequency"]
on0 = rng.integers(0, 256, N)
idx = np.arange(N)
gate = plans["gate"]


ONLY = sys.argv[1] if len(sys.argv) > 1 else None


def run(name, ev, graph="fm_voice", setfreq=None):
    if ONLY and not name.startswith(ONLY):
        return
    e = oscen_amd.Engine(graph, N, sample_rate=48000.0)
    e.set_voice_values("frequency", freq)
    for inp, (v, f, x) in ev:
        o = np.lexsort((f, v))
        e.schedule_voice_events(inp, v[o], f[o], x[o])
    e.set_bus_batching(8)
    for _ in range(W):
        e.process_block_async(BLOCK)
    e.flush(); e.synchronize()
    e.enable_kernel_timing(True)
    for _ in range(K):
        e.process_block_async(BLOCK)
    e.flush(); e.synchronize()
    ms = e.kernel_time_ms()
    print("%-34s kernel ms/block %.5f" % (name, ms[0] * ms[1] / K if isinstance(ms, tuple) else ms), flush=True)


def cat(*parts):
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))


note_on = (idx, on0, gate)
t0, t1 = W * BLOCK, (W + K) * BLOCK
rnd = lambda: rng.integers(t0, t1, N)
run("W0 note-on only", [("gate", note_on)])
run("W1 + 1 retrigger/voice", [("gate", cat(note_on, (idx, rnd(), gate)))])
half = idx[::2]
run("W1h + 1 retrigger / 2 voices", [("gate", cat(note_on, (half, rnd()[::2], gate[::2])))])
run("W3 all in release", [("gate", cat(note_on, (idx, 600 + rng.integers(0, 256, N), np.zeros(N, np.float32))))])
run("W3b + note-off in timed region", [("gate", cat(note_on, (idx, rnd(), np.zeros(N, np.float32))))])
pl = oscen_amd.note_plans(N, span=(W + K) * BLOCK, fold="slice")
v, f, x = pl["events"]
keep = f < t1
run("W4 slice plan (driver)", [("gate", (v[keep], f[keep], x[keep]))])
# the same events, but each wave's 64 voices share their timing (events of voice 64k+i moved onto the frames of voice 64k)
run("W5 idle (no events at all)", [])
run("W6 + 1 set-frequency/voice (same f)", [("gate", note_on), ("frequency", (idx, rnd(), freq))])
run("W7 + 2 retriggers/voice", [("gate", cat(note_on, (idx, rnd(), gate), (idx, rnd(), gate)))])
run("W8 note-on velocity 0 -> stays idle + 1 noteoff/voice (idle->release?)", [("gate", (idx, rnd(), np.zeros(N, np.float32)))])
