// How accurate -- and how much cheaper -- is the hardware sine on gfx950?  (the polynomial sine is 12
// of the ~24 VALU instructions an FM operator spends per frame; v_sin_f32 takes its argument in TURNS, which is what an
// operator's phase is.)  Prints, for x = fl32(t * 2 pi) with t the operator's phase + modulation in turns:
//   * the error of og_sinf (the shipped polynomial, og_math.h) against sin(double(x)),
//   * the error of v_sin_f32(fract(x * (1 / 2 pi))) and of v_sin_f32(fract(t)) against the same,
//   * ns per wave-instruction-equivalent of both in a dependent chain (one FM operator's worth of work per step).
//   hipcc --offload-arch=gfx950 -O2 -I oscen_amd/csrc -o /tmp/vsin scripts/ubench/vsin.hip && /tmp/vsin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "og_math.h"

__device__ __forceinline__ float hw_sin_turns(float t) { return __builtin_amdgcn_sinf(t - floorf(t)); }

__global__ void eval(const float* t_in, float* poly, float* hw_x, float* hw_t, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = t_in[i];
    const float x = t * 6.28318548202514648f; // the reference's f32 product: part of the contract
    poly[i] = og_sinf(x);
    hw_x[i] = hw_sin_turns(x * 0.15915494309189535f);
    hw_t[i] = hw_sin_turns(t);
}

template <int MODE>
__global__ void __launch_bounds__(64) chain(float* out, int iters)
{
    float phase = threadIdx.x * 0.001f, prev = 0.0f, acc = 0.0f;
    const float inc = 0.0091f, fb = 0.3f;
    for (int i = 0; i < iters; ++i) { // one FM operator per trip: the sine depends on the previous output
        const float t = phase + prev * fb;
        const float s = MODE == 0 ? og_sinf(t * 6.28318548202514648f) : hw_sin_turns(t);
        prev = s * 0.7f;
        acc += prev;
        phase += inc;
        phase -= truncf(phase);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> t(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { // phases in [0, 1) plus modulation of a few turns, both signs
        s = s * 1664525u + 1013904223u;
        const float u = (s >> 8) * (1.0f / 16777216.0f);
        s = s * 1664525u + 1013904223u;
        const float m = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * (i % 4 == 0 ? 0.0f : (i % 4 == 1 ? 1.0f : (i % 4 == 2 ? 4.0f : 16.0f)));
        t[i] = u + m;
    }
    float *d_t, *d_a, *d_b, *d_c;
    hipMalloc(&d_t, n * 4); hipMalloc(&d_a, n * 4); hipMalloc(&d_b, n * 4); hipMalloc(&d_c, n * 4);
    hipMemcpy(d_t, t.data(), n * 4, hipMemcpyHostToDevice);
    eval<<<(n + 255) / 256, 256>>>(d_t, d_a, d_b, d_c, n);
    std::vector<float> a(n), b(n), c(n);
    hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d_c, n * 4, hipMemcpyDeviceToHost);
    const char* names[3] = {"og_sinf(fl32(t * 2pi)) [shipped]", "v_sin_f32(fract(x / 2pi))", "v_sin_f32(fract(t))"};
    const std::vector<float>* got[3] = {&a, &b, &c};
    for (int k = 0; k < 3; ++k) {
        double worst[4] = {0, 0, 0, 0}, sum2 = 0.0;
        for (int i = 0; i < n; ++i) {
            const float x = t[i] * 6.28318548202514648f;
            const double ref = std::sin((double)x);
            const double e = std::fabs((double)(*got[k])[i] - ref);
            if (e > worst[i % 4]) worst[i % 4] = e;
            sum2 += e * e;
        }
        printf("%-34s max abs error by modulation depth 0 / 1 / 4 / 16 turns: %.3g %.3g %.3g %.3g   rms %.3g\n", names[k], worst[0], worst[1], worst[2], worst[3],
               std::sqrt(sum2 / n));
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float* d_out;
    hipMalloc(&d_out, 1024 * 8 * 64 * 4);
    for (int mode = 0; mode < 2; ++mode)
        for (int waves : {1, 4}) {
            const int iters = 100000, blocks = 1024 * waves;
            if (mode == 0) chain<0><<<blocks, 64>>>(d_out, 1000); else chain<1><<<blocks, 64>>>(d_out, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) chain<0><<<blocks, 64>>>(d_out, iters); else chain<1><<<blocks, 64>>>(d_out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s, %d waves per SIMD: %.2f ns per operator step per SIMD\n", mode == 0 ? "polynomial sine" : "hardware sine  ", waves,
                   ms * 1e6 / ((double)iters * waves));
        }
    return 0;
}
