"""Where does a 20-30 ms block of the blocking entry at 4 M voices come from?  (Seen once per ~15 000-30 000 blocks,
back to back at full load: bench.py's realtime record.)

    python scripts/dbg_rt_hiccup.py [voices] [blocks] [paced|back] [loaded]

`loaded`: the synthetic score of the throughput lines (every voice plays its cyclic 1 s note plan) is resident for the
whole run, on top of the live messages -- the loaded bank of bench.py's real-time record.

Windows of 50 blocks; per window the host latency of every block and the kernel time the engine's HIP events saw.
A window with a slow block prints both: kernel time up by the same amount = the GPU ran the kernel slowly (clock /
power event); kernel time unchanged = the time went by outside the kernel (submission, host).  `paced` sleeps to the
5.33 ms block deadline between calls (what a sound card does) instead of calling back to back."""
import sys
import time

import numpy as np

import oscen_amd

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
paced = len(sys.argv) > 3 and sys.argv[3] == "paced"
loaded = len(sys.argv) > 4 and sys.argv[4] == "loaded"
block, W = 256, 50
eng = oscen_amd.Engine("fm_voice", V, sample_rate=48000.0)
if loaded:
    plans = oscen_amd.note_plans(V)
    total_frames = (50 + N + 2) * block
    ev_v, ev_f0, ev_x = plans["events"]
    reps = -(-total_frames // 48000)
    plans["events"] = (np.tile(ev_v, reps), np.concatenate([ev_f0 + 48000 * k for k in range(reps)]), np.tile(ev_x, reps))
    eng.reserve_events(int(1000 * (50 + N) * 3.0 * total_frames / 48000.0) + (1 << 20))
    print("resident score events:", oscen_amd.schedule_note_plans(eng, plans, total_frames=total_frames), flush=True)
    del plans
else:
    eng.set_voice_values("frequency", oscen_amd.note_plans(V)["frequency"])
midi = oscen_amd.Midi(eng)
midi.set_queue_capacity(1000)
rng = np.random.default_rng(1)
notes = rng.integers(36, 97, size=(64, 1000)).astype(np.uint8)
frames = np.sort(rng.integers(0, block, size=(64, 1000)), axis=1).astype(np.uint32)
packed = [midi.pack_messages(notes[i - (i % 2)], frames[i], on=(i % 2 == 0)) for i in range(64)]
for i in range(50):
    midi.send_packed(packed[i % 64])
    midi.process_block(block)
eng.enable_kernel_timing(True)
deadline = block / 48000.0
base = []
slow = 0
t_next = time.perf_counter()
for w in range(N // W):
    lat = np.empty(W)
    eng.kernel_time_ms()  # (reading resets the window)
    for i in range(W):
        if paced:
            t_next += deadline
            while time.perf_counter() < t_next:
                pass
        t0 = time.perf_counter()
        midi.send_packed(packed[(w * W + i) % 64])
        midi.process_block(block)
        lat[i] = time.perf_counter() - t0
    k_ms, n = eng.kernel_time_ms()
    k_sum = k_ms * n
    if len(base) < 20:
        base.append((lat.sum() * 1e3, k_sum))
    if lat.max() * 1e3 > 4.0:
        slow += 1
        b = np.median(np.array(base), axis=0)
        print(f"window {w}: worst block {lat.max() * 1e3:.2f} ms at {int(lat.argmax())}; window host {lat.sum() * 1e3:.2f} ms "
              f"(typical {b[0]:.2f}), kernel events {k_sum:.2f} ms over {n} launches (typical {b[1]:.2f})", flush=True)
b = np.median(np.array(base), axis=0)
print(f"{V} voices, {N} blocks, paced={paced}: {slow} slow windows; typical window host {b[0]:.2f} ms kernel {b[1]:.2f} ms; "
      f"blocking stats {eng.blocking_stats}")
