"""The entry points only bench.py used to call -- og_process_block_async into a device buffer, og_flush, og_synchronize,
og_enable_kernel_timing / og_kernel_time_ms / og_kernel_blocks_timed, og_reserve_events -- at a size a test can afford: the
queued path must give the blocking path's bus bit for bit and the counters must add up.  (`-m gpu`; the host simulator runs
it too: the line-coverage run of round 4 showed these as the largest untested part of the engine.)"""
import ctypes as C

import numpy as np
import pytest

import oscen_amd

pytestmark = pytest.mark.gpu
SR = 48000.0


class DeviceBuffer:
    """device memory through the HIP runtime the loaded library itself is linked against (dlsym on its handle also searches
    its dependencies), so the same test runs against liboscen_gpu.so and against the host simulator"""

    def __init__(self, nbytes):
        self.rt = oscen_amd.load_library()
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        assert self.rt.hipMalloc(C.byref(self.ptr), nbytes) == 0
        assert self.rt.hipMemset(self.ptr, 0, nbytes) == 0

    def to_host(self):
        out = np.empty(self.nbytes // 4, dtype=np.float32)
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            self.rt.hipFree(self.ptr)
            self.ptr = None


@pytest.mark.parametrize("batch", [1, 4, 0])
def test_queued_blocks_into_a_device_buffer_equal_the_blocking_blocks_and_the_timers_count_them(batch):
    n, block, nb = 300, 256, 10
    total = block * nb
    plans = oscen_amd.note_plans(n, span=total, fold="slice")
    ref = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    oscen_amd.schedule_note_plans(ref, plans, total_frames=total)
    want = np.concatenate([ref.process_block(block).copy() for _ in range(nb)], axis=0)

    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    eng.reserve_events(50000)  # (before the score goes to the device: room behind it for live segments)
    oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
    if batch != 1:
        eng.set_bus_batching(batch)
    ch = eng.channels
    buf = DeviceBuffer(nb * block * ch * 4)
    try:
        eng.enable_kernel_timing(True)
        for i in range(nb):
            if i == 6:  # a live push in the middle: the queue is launched first, the event joins the timeline behind the score
                eng.push_voice_event("gate", 7, 11, 0.9)
            eng.process_block_async(block, buf.ptr.value + i * block * ch * 4)
        eng.flush()
        eng.synchronize()
        ms, launches = eng.kernel_time_ms()
        assert eng.kernel_blocks_timed == nb and 1 <= launches <= nb and ms >= 0.0
        if batch == 4:
            assert launches <= 4  # 6 blocks before the push = 4 + 2, then 4 = one more launch
        got = buf.to_host().reshape(nb * block, ch)
    finally:
        buf.free()
    # the same run on the blocking entry, with the same live push
    ref = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    oscen_amd.schedule_note_plans(ref, plans, total_frames=total)
    rows = []
    for i in range(nb):
        if i == 6:
            ref.push_voice_event("gate", 7, 11, 0.9)
        rows.append(ref.process_block(block).copy())
    assert np.array_equal(got, np.concatenate(rows, axis=0))
    assert not np.array_equal(got, want)  # (the push is audible)
    assert eng.event_stats["full_rebuilds"] >= 1
    with pytest.raises(oscen_amd.OscenError):
        eng.reserve_events(1 << 33)


def test_the_kernel_reads_its_own_clock_and_the_probe_reads_the_idle_one():
    """og_kernel_clock_ghz: workgroup 0 of every TIMED launch writes {shader cycles, 100 MHz ticks} at its first and last
    instruction; og_shader_clock_ghz: a 20 us one-wave probe.  On the MI355X both land between the part's idle and boost
    clocks; the host simulator has no such counters (0 / None).  Timing off: no marks are written, the getter says 0."""
    n, frames = 4096, 256
    eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
    eng.set_voice_values("frequency", oscen_amd.note_plans(n)["frequency"])
    eng.process_block(frames)
    assert eng.kernel_clock_ghz == 0.0                      # nothing timed yet
    eng.enable_kernel_timing(True)
    for _ in range(6):
        eng.process_block(frames)
    ms, launches = eng.kernel_time_ms()
    assert launches == 6 and ms > 0.0
    k = eng.kernel_clock_ghz
    p = eng.shader_clock_ghz()
    if p is None:                                           # the simulator: no shader clock
        assert k == 0.0
    else:
        assert 0.3 < k < 3.5 and 0.3 < p < 3.5, (k, p)
        # cycles between the marks / launch duration agree with the clock (the marks bracket workgroup 0, the events the launch)
        assert abs(k - p) < 1.5
    eng.enable_kernel_timing(False)
    eng.process_block(frames)
    assert eng.kernel_time_ms()[0] < 0.0                    # timing off


@pytest.mark.parametrize("batch", [1, 0])
def test_blocks_handed_over_in_one_call_equal_the_blocks_handed_over_one_by_one(batch):
    """og_process_blocks_async (what bench.py's timed region calls where nothing happens between its blocks): n blocks in
    one crossing of the boundary ≡ n og_process_block_async calls, bit for bit, buses at base + b * stride; a stride
    wider than a block leaves the gap untouched; bad arguments are refused."""
    n, block, nb = 200, 256, 9
    total = block * nb
    plans = oscen_amd.note_plans(n, span=total, fold="slice")

    def run(one_call, stride_blocks=1):
        eng = oscen_amd.Engine("fm_voice", n, sample_rate=SR)
        oscen_amd.schedule_note_plans(eng, plans, total_frames=total)
        if batch != 1:
            eng.set_bus_batching(batch)
        ch = eng.channels
        stride = stride_blocks * block * ch * 4
        buf = DeviceBuffer(nb * stride)
        try:
            if one_call:
                eng.process_blocks_async(block, 4, buf.ptr.value, stride)            # two calls: 4 + 5 blocks
                eng.process_blocks_async(block, nb - 4, buf.ptr.value + 4 * stride, stride)
            else:
                for i in range(nb):
                    eng.process_block_async(block, buf.ptr.value + i * stride)
            eng.synchronize()
            return buf.to_host().reshape(nb, stride_blocks, block * ch), eng
        finally:
            buf.free()

    a, _ = run(False)
    b, eng = run(True)
    assert np.array_equal(a, b) and np.abs(a).max() > 0.0
    c, _ = run(True, stride_blocks=2)
    assert np.array_equal(c[:, 0], a[:, 0]) and not c[:, 1].any()
    with pytest.raises(oscen_amd.OscenError):
        eng.process_blocks_async(0, 3)
    with pytest.raises(oscen_amd.OscenError):
        eng.process_blocks_async(513, 3)
    eng.process_blocks_async(block, 0)  # nothing to do
