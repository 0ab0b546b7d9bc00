// Does a wave64 VALU instruction with only 32 active lanes issue faster on gfx950?  (If it did, 32 voices per wave
// would buy twice the waves per SIMD at the 65 536-voice headline for the same issue cycles.)
//   hipcc --offload-arch=gfx950 -O2 -o halfwave halfwave.hip && ./halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ACTIVE>
__global__ void __launch_bounds__(64) chain(float* out, int iters)
{
    const int lane = threadIdx.x;
    float a = lane * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
    if (lane < ACTIVE) {
        for (int i = 0; i < iters; ++i) { // 8 dependent-free FMAs per trip (4 chains x 2)
            a = __builtin_fmaf(a, b, 0.001f);
            c = __builtin_fmaf(c, b, 0.002f);
            d = __builtin_fmaf(d, b, 0.003f);
            b = __builtin_fmaf(b, 0.99999f, 0.00001f);
            a = __builtin_fmaf(a, 0.9999f, c);
            c = __builtin_fmaf(c, 0.9999f, d);
            d = __builtin_fmaf(d, 0.9999f, a);
            b = __builtin_fmaf(b, 0.99999f, 0.00001f);
        }
        out[blockIdx.x * 64 + lane] = a + b + c + d;
    }
}

template <int ACTIVE>
static void run(int waves_per_simd, float* d_out)
{
    const int iters = 200000, blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    chain<ACTIVE><<<blocks, 64>>>(d_out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<ACTIVE><<<blocks, 64>>>(d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = 8.0 * iters * waves_per_simd; // VALU instructions per SIMD
    printf("active lanes %2d, %d waves per SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", ACTIVE,
           waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
}

int main()
{
    float* d_out;
    hipMalloc(&d_out, 1024 * 16 * 64 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<64>(w, d_out);
        run<32>(w, d_out);
        run<16>(w, d_out);
    }
    return 0;
}
