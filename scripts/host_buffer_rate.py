"""PCIe-inclusive rate of the headline configuration: the caller takes every block's bus
into HOST memory with the blocking og_process_block (one 1 KB D2H copy + a stream sync per block)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oscen_amd  # noqa: E402

V, block, K, W = 65536, 256, 188, 8
eng = oscen_amd.Engine("fm_voice", V, sample_rate=48000.0)
oscen_amd.schedule_note_plans(eng, oscen_amd.note_plans(V), total_frames=(K + W) * block)
for _ in range(W):
    eng.process_block(block)
t0 = time.perf_counter()
for _ in range(K):
    eng.process_block(block)
dt = time.perf_counter() - t0
print("host-buffer (blocking) path: %.4g voices*samples/s, %.4f ms per block" % (V * K * block / dt, dt / K * 1e3))
