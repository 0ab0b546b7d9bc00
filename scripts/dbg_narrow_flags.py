"""A/B helper (round 6): the bus of a bank rendered with the library OSCEN_GPU_LIB names, written to a file -- two libraries that
differ in the hand-off only must give the same bytes.  usage: python scripts/dbg_narrow_flags.py <voices> <out.npy>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oscen_amd

n, out = int(sys.argv[1]), sys.argv[2]
blocks, block = 12, 256
e = oscen_amd.Engine("fm_voice", n, sample_rate=48000.0)
plans = oscen_amd.note_plans(n, span=blocks * block, fold="cyclic")
oscen_amd.schedule_note_plans(e, plans, total_frames=blocks * block)
e.set_bus_batching(4)
bus = [e.process_block(block) for _ in range(blocks)]
print(e.kernel_variant, e.pipeline_depth, float(np.abs(np.concatenate(bus)).max()))
np.save(out, np.concatenate(bus))
