"""The first launch of the voice kernel after FOREIGN kernels is ~35 % slower.  Two 20-block launches of fm_voice
(65 536 voices): one right after another voice-kernel launch, one right after a burst of torch element-wise kernels.
Run under rocprofv3 --pmc to compare the two dispatches (scripts/dbg_first_launch.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oscen_amd

eng = oscen_amd.Engine("fm_voice", 65536, sample_rate=48000.0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_bus_batching(32)
foreign = sys.argv[1] if len(sys.argv) > 1 else "torch"
x = torch.empty(1 << 22, dtype=torch.float32, device="cuda")


def launch(tag):
    eng.enable_kernel_timing(True)
    for _ in range(20):
        eng.process_block_async(256)
    eng.flush(); eng.synchronize()
    ms, n = eng.kernel_time_ms()
    eng.enable_kernel_timing(False)
    print("%-40s %.4f ms" % (tag, ms), flush=True)


launch("warm-up")
launch("A: after a voice-kernel launch")
if foreign == "torch":
    for _ in range(16):
        x.mul_(1.0001).add_(1.0)
elif foreign == "sleep":
    import time; time.sleep(0.05)
elif foreign == "small":  # one tiny foreign kernel
    x[:64].add_(1.0)
torch.cuda.synchronize()
launch("B: after foreign work (%s)" % foreign)
launch("C: the launch after B")
