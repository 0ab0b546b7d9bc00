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
__device__ __forceinline__ void step(float (&x)[CH], double (&xp)[CH], float a, float b, double ap, double bp)
{
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (OP == 2) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
        if (OP == 3) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
        if (OP == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(a) : );
        if (OP == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        // compare + select pairs as the compiler writes them (it puts `s_nop 1` between a VALU write of the mask and the select)
        if (OP == 18) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
        if (OP == 19) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(x[i]) : "v"(a), "v"(b) : "s20", "s21");
        if (OP == 20) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
        if (OP == 15) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (OP == 16) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
        if (OP == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
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
        if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[i]) : "v"(ap), "v"(bp));
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
    double xp[CH]; // eight independent 64-bit register pairs for the packed form
    union { float f[2]; double d; } ua = {{a, a}}, ub = {{b, b}};
#pragma unroll
    for (int i = 0; i < CH; ++i) { union { float f[2]; double d; } u = {{x[i], x[i] * 0.5f}}; xp[i] = u.d; }
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555" ::: "vcc", "s20", "s21");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) step<OP>(x, xp, a, b, ua.d, ub.d);
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) { union { double d; float f[2]; } u; u.d = xp[i]; s += x[i] + u.f[0] + u.f[1]; }
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
    run<3>("v_cndmask_b32 (vcc)", out, cyc, n_simd);
    run<13>("v_cndmask_b32 (sgpr)", out, cyc, n_simd);
    run<14>("v_max_f32", out, cyc, n_simd);
    run<20>("v_cmp_gt_f32 -> vcc", out, cyc, n_simd);
    run<18>("cmp+nop+cndmask vcc", out, cyc, n_simd);
    run<19>("cmp+nop+cndmask sgpr", out, cyc, n_simd);
    run<15>("v_med3_f32", out, cyc, n_simd);
    run<16>("v_cvt_f32_u32", out, cyc, n_simd);
    run<17>("v_add_u32", out, cyc, n_simd);
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
