// Probe: is packed f32 (v_pk_*: two voices per lane) really twice the arithmetic per issue slot on gfx950?
// Same FM-operator recurrence, once on float (64 voices per wave) and once on float2 (128 per wave).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_probe.hip -o pk_probe && ./pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float F2 __attribute__((ext_vector_type(2)));
typedef unsigned U2 __attribute__((ext_vector_type(2)));
template <class T> struct V;
template <> struct V<float> {
    using U = unsigned;
    static __device__ float splat(float x) { return x; }
    static __device__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ float trunc(float x) { return __builtin_truncf(x); }
};
template <> struct V<F2> {
    using U = U2;
    static __device__ F2 splat(float x) { return (F2){x, x}; }
    static __device__ F2 fma(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ F2 trunc(F2 x) { return __builtin_elementwise_trunc(x); }
};
template <class T>
__device__ __forceinline__ T sinx(T x)
{
    using W = V<T>;
    const T k = W::fma(x, W::splat(0x1.45f306p-2f), W::splat(12582912.0f));
    const T n = k - W::splat(12582912.0f);
    T r = W::fma(n, W::splat(-0x1.921fb6p+1f), x);
    r = W::fma(n, W::splat(0x1.777a5cp-24f), r);
    typename W::U sgn = __builtin_bit_cast(typename W::U, k) << 31;
    r = __builtin_bit_cast(T, __builtin_bit_cast(typename W::U, r) ^ sgn);
    const T s = r * r;
    T p = W::fma(s, W::splat(-0x1.9d0bc6p-26f), W::splat(0x1.6fadb6p-19f));
    p = W::fma(s, p, W::splat(-0x1.a018e8p-13f));
    p = W::fma(s, p, W::splat(0x1.111110p-7f));
    p = W::fma(s, p, W::splat(-0x1.555556p-3f));
    return W::fma(s * r, p, r);
}
template <class T>
__device__ __forceinline__ T run(T phase, T inc, int frames)
{
    using W = V<T>;
    T prev = W::splat(0.f), acc = W::splat(0.f), p2 = phase * W::splat(0.5f), prev2 = W::splat(0.f);
    for (int f = 0; f < frames; ++f) { // two chained operators per frame
        const T o = sinx<T>((phase + prev * W::splat(0.3f)) * W::splat(6.28318548202514648f)) * W::splat(0.7f);
        prev = o;
        const T o2 = sinx<T>((p2 + o) * W::splat(6.28318548202514648f)) * W::splat(0.9f);
        prev2 = o2;
        T p = phase + inc;
        phase = p - W::trunc(p);
        p = p2 + inc;
        p2 = p - W::trunc(p);
        acc += o2;
    }
    return acc + prev2;
}
__global__ __launch_bounds__(64) void k1(float* out, const float* in, int frames)
{
    const int v = blockIdx.x * 64 + threadIdx.x;
    out[v] = run<float>(in[v], in[v] * 0.01f + 0.003f, frames);
}
__global__ __launch_bounds__(64) void k2(float* out, const float* in, int frames)
{
    const int v = blockIdx.x * 128 + threadIdx.x;
    const F2 r = run<F2>((F2){in[v], in[v + 64]}, (F2){in[v] * 0.01f + 0.003f, in[v + 64] * 0.01f + 0.003f}, frames);
    out[v] = r.x;
    out[v + 64] = r.y;
}
int main()
{
    const int frames = 4096;
    for (int V : {65536, 131072, 262144, 524288}) {
        float *in, *o1, *o2;
        hipMalloc(&in, V * 4); hipMalloc(&o1, V * 4); hipMalloc(&o2, V * 4);
        std::vector<float> h(V);
        for (int i = 0; i < V; ++i) h[i] = (i % 997) / 997.0f;
        hipMemcpy(in, h.data(), V * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms1 = 0, ms2 = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a); hipLaunchKernelGGL(k1, dim3(V / 64), dim3(64), 0, 0, o1, in, frames); hipEventRecord(b);
            hipEventSynchronize(b); hipEventElapsedTime(&ms1, a, b);
            hipEventRecord(a); hipLaunchKernelGGL(k2, dim3(V / 128), dim3(64), 0, 0, o2, in, frames); hipEventRecord(b);
            hipEventSynchronize(b); hipEventElapsedTime(&ms2, a, b);
        }
        std::vector<float> r1(V), r2(V);
        hipMemcpy(r1.data(), o1, V * 4, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), o2, V * 4, hipMemcpyDeviceToHost);
        int diff = 0;
        for (int i = 0; i < V; ++i) diff += r1[i] != r2[i];
        printf("V=%d scalar %.3f ms packed %.3f ms speedup %.2f bit-diffs %d (waves/SIMD scalar %.1f packed %.1f)\n", V, ms1, ms2, ms1 / ms2, diff,
               V / 64 / 1024.0, V / 128 / 1024.0);
        hipFree(in); hipFree(o1); hipFree(o2);
    }
    return 0;
}
