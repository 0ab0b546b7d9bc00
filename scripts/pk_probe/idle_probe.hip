// Probe: what a kernel launch costs after the GPU has been idle, against the same launch in a stream of launches.
// A 1024-workgroup kernel that spins for a fixed number of shader cycles (~100 us); per launch: the HIP-event time around
// it (hipEventRecord before and after, as og_engine.cpp times the voice kernel) and the host's wall time from submission to
// completion.     hipcc --offload-arch=gfx950 -O3 -o idle_probe idle_probe.hip && ./idle_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void spin(unsigned long long cycles, unsigned long long* out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = __builtin_amdgcn_s_memtime() - t0;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    unsigned long long* out; (void)hipMalloc(&out, 64);
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t a[64], b[64];
    for (int i = 0; i < 64; ++i) { (void)hipEventCreate(&a[i]); (void)hipEventCreate(&b[i]); }
    const unsigned long long cyc = 240000; // ~100 us at 2.4 GHz
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, cyc, out);
    (void)hipStreamSynchronize(s);
    // (a) a stream of 16 launches, no synchronisation between them
    const double w0 = now_us();
    for (int i = 0; i < 16; ++i) { (void)hipEventRecord(a[i], s); hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, cyc, out); (void)hipEventRecord(b[i], s); }
    (void)hipStreamSynchronize(s);
    const double w1 = now_us();
    std::printf("stream of 16: wall %.1f us per launch; event times:", (w1 - w0) / 16);
    for (int i = 0; i < 16; ++i) { float ms; (void)hipEventElapsedTime(&ms, a[i], b[i]); std::printf(" %.1f", ms * 1e3); }
    std::printf("\n");
    // (b) one launch after the GPU has been idle for d microseconds
    for (int d : {0, 20, 50, 100, 200, 500, 1000, 5000, 20000}) {
        std::vector<double> ev, wall;
        for (int rep = 0; rep < 12; ++rep) {
            (void)hipStreamSynchronize(s);
            if (d) { const double t = now_us(); while (now_us() - t < d) {} }
            const double t0 = now_us();
            (void)hipEventRecord(a[0], s);
            hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, cyc, out);
            (void)hipEventRecord(b[0], s);
            (void)hipStreamSynchronize(s);
            const double t1 = now_us();
            float ms; (void)hipEventElapsedTime(&ms, a[0], b[0]);
            ev.push_back(ms * 1e3); wall.push_back(t1 - t0);
        }
        std::sort(ev.begin(), ev.end()); std::sort(wall.begin(), wall.end());
        std::printf("idle %6d us, then one launch: event time median %.1f us (min %.1f, max %.1f); submit -> synchronised median %.1f us\n", d, ev[6], ev[0], ev[11], wall[6]);
    }
    return 0;
}
