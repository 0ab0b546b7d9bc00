// Probe: issue cost (SIMD cycles per wave64 instruction) of the instruction classes the voice kernels are made of, at
// 1 / 2 / 4 / 8 waves per SIMD, eight independent chains per wave (throughput, not latency), timed with s_memtime inside
// the wave.  Classes: plain f32 (v_fma_f32, v_mul_f32, v_fract_f32, v_cndmask_b32), transcendental (v_rcp_f32,
// v_sin_f32, v_exp_f32), and mixes (3 fma : 1 rcp, 7 fma : 1 rcp) that show whether a transcendental overlaps with the
// plain stream or holds the pipe.     hipcc --offload-arch=gfx950 -O3 -o trans_probe trans_probe.hip && ./trans_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CH 8
#define ITERS 2048

template <int OP>
__device__ __forceinline__ void step(float (&x)[CH], float a, float b)
{
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (OP == 2) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
        if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
        if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        if (OP == 5) asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]));
        if (OP == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        if (OP == 7) { // 3 fma : 1 rcp
            if (i % 4 == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
        if (OP == 8) { // 7 fma : 1 rcp
            if (i == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
        if (OP == 9) { // 7 fma : 1 sin
            if (i == 7) asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
        if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&x[i & ~1]) : "v"(*(double*)&a), "v"(*(double*)&b));
        if (OP == 11) asm volatile("s_mul_i32 s20, s20, s21" ::: "s20");
        if (OP == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    }
}

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float a, float b)
{
    float x[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = 0.5f + 0.01f * (float)(threadIdx.x + i);
    float aa[2] = {a, a}, bb[2] = {b, b};
    (void)aa; (void)bb;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) step<OP>(x, a, b);
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, float* out, unsigned long long* cyc, int n_simd)
{
    std::printf("%-22s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = n_simd * w;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, 0.999f, 0.001f); // warm
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, 0.999f, 0.001f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[grid / 2];
        const double per = med / ((double)ITERS * CH * w); // SIMD cycles per wave-instruction (s_memtime ticks: 100 MHz? see ratio)
        std::printf("  w%d: %7.3f tick/inst (%6.1f us)", w, per * 1.0, ms * 1e3);
    }
    std::printf("\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_simd = p.multiProcessorCount * 4;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)n_simd * 8 * 64 * 4); hipMalloc(&cyc, (size_t)n_simd * 8 * 8);
    std::printf("CUs %d, clock %d kHz; per class: s_memtime ticks per wave64 instruction per SIMD (and the launch's wall time; instructions per wave = %d)\n",
                p.multiProcessorCount, p.clockRate, ITERS * CH);
    run<0>("v_fma_f32", out, cyc, n_simd);
    run<1>("v_mul_f32", out, cyc, n_simd);
    run<12>("v_sub_f32", out, cyc, n_simd);
    run<2>("v_fract_f32", out, cyc, n_simd);
    run<3>("v_cndmask_b32", out, cyc, n_simd);
    run<10>("v_pk_fma_f32", out, cyc, n_simd);
    run<4>("v_rcp_f32", out, cyc, n_simd);
    run<5>("v_sin_f32", out, cyc, n_simd);
    run<6>("v_exp_f32", out, cyc, n_simd);
    run<7>("3 fma : 1 rcp", out, cyc, n_simd);
    run<8>("7 fma : 1 rcp", out, cyc, n_simd);
    run<9>("7 fma : 1 sin", out, cyc, n_simd);
    run<11>("s_mul_i32", out, cyc, n_simd);
    return 0;
}
