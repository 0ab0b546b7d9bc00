"""Why is the voice kernel ~35 % slower once an RCCL communicator exists in the process?  Times one 20-block launch of
fm_voice (65 536 voices) after each step."""
import ctypes as C, os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oscen_amd

torch.cuda.set_device(0)
eng = oscen_amd.Engine("fm_voice", 65536, sample_rate=48000.0)
plans = oscen_amd.note_plans(65536, span=6400, fold="slice")
oscen_amd.schedule_note_plans(eng, plans, total_frames=10**9)
eng.set_bus_batching(32)
if os.environ.get("DBG_TORCH_STREAM"):
    eng.set_stream(torch.cuda.current_stream().cuda_stream)


def measure(tag, warm=True):
    for _ in range(5 if warm else 0):
        eng.process_block_async(256)
    eng.flush(); eng.synchronize()
    eng.enable_kernel_timing(True)
    for _ in range(20):
        eng.process_block_async(256)
    eng.flush(); eng.synchronize()
    ms, n = eng.kernel_time_ms()
    eng.enable_kernel_timing(False)
    print("%-44s %.4f ms per launch (%d launches) = %.5f ms/block" % (tag, ms, n, ms * n / 20.0), flush=True)


measure("baseline")
hip = C.CDLL("libamdhip64.so")
val = C.c_size_t()
for name, lim in (("hipLimitStackSize", 0), ("hipLimitMallocHeapSize", 2)):
    hip.hipDeviceGetLimit(C.byref(val), lim)
    print(name, val.value)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("stack", "all"):
    print("hipDeviceSetLimit(stack, 16384) ->", hip.hipDeviceSetLimit(0, C.c_size_t(16384)))
    measure("after hipDeviceSetLimit(stack 16 KB)")
if which in ("rccl", "all"):
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    print("ncclCommInitAll ->", rccl.ncclCommInitAll(C.byref(comm), 1, devs))
    measure("after ncclCommInitAll(1 device)")
    hip.hipDeviceGetLimit(C.byref(val), 0)
    print("stack limit now", val.value)
    rccl.ncclCommDestroy(comm)
    measure("after ncclCommDestroy")
if which in ("torch", "all"):
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    measure("after init_process_group (no collective yet)")
    t = torch.ones(1024, device="cuda")
    dist.all_reduce(t); torch.cuda.synchronize()
    measure("after the first all_reduce")
    dist.barrier(device_ids=[0]); torch.cuda.synchronize()
    measure("after a barrier")
    bus = torch.zeros((25, 256), dtype=torch.float32, device="cuda")
    dist.reduce(bus[:5], dst=0); torch.cuda.synchronize()
    measure("after dist.reduce of a bus tensor")
    for k in range(3):
        dist.reduce(bus[:5], dst=0); torch.cuda.synchronize()
        dist.barrier(device_ids=[0]); torch.cuda.synchronize()
        measure("first launch right after reduce+barrier #%d" % k, warm=False)
        measure("  and the launch after that", warm=False)
    import time
    time.sleep(0.05)
    measure("first launch after 50 ms of idling", warm=False)
    measure("  and the launch after that", warm=False)
    dist.destroy_process_group()
    measure("after destroy_process_group")
