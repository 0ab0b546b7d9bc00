// Probe for VERDICT r1 item 5: what would a packed-f32 (two voices per lane) variant of the WHOLE fm_voice tick buy?
// The sustain/release-path arithmetic of one FMVoice frame -- 4 ADSR ticks with the release division, 3 chained FM
// operators, the TPT core, gain -- written once over T = float (64 voices per wave) and T = float2 (128 per wave,
// v_pk_* where an instruction has a packed form; cvt / rcp / integer ops run once per half).  State lives in
// registers over the frame loop like in the real kernel (30 words per voice).  Timing only: not the product path.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize voice_probe.hip -o voice_probe && ./voice_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float F2 __attribute__((ext_vector_type(2)));
typedef unsigned U2 __attribute__((ext_vector_type(2)));
template <class T> struct V;
template <> struct V<float> {
    using U = unsigned;
    static __device__ float splat(float x) { return x; }
    static __device__ unsigned usplat(unsigned x) { return x; }
    static __device__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ float trunc(float x) { return __builtin_truncf(x); }
    static __device__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ float u2f(unsigned u) { return (float)u; }
};
template <> struct V<F2> {
    using U = U2;
    static __device__ F2 splat(float x) { return (F2){x, x}; }
    static __device__ U2 usplat(unsigned x) { return (U2){x, x}; }
    static __device__ F2 fma(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ F2 trunc(F2 x) { return __builtin_elementwise_trunc(x); }
    static __device__ F2 rcp(F2 x) { return (F2){__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
    static __device__ F2 u2f(U2 u) { return (F2){(float)u.x, (float)u.y}; }
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
    T p = W::fma(s, W::splat(-0x1.9d0bc6p-26f), W::splat(0x1.7190e0p-19f));
    p = W::fma(s, p, W::splat(-0x1.a018e8p-13f));
    p = W::fma(s, p, W::splat(0x1.111110p-7f));
    p = W::fma(s, p, W::splat(-0x1.555556p-3f));
    return W::fma(s * r, p, r);
}
template <class T> struct Env { T lv, tgt, cf, rs; typename V<T>::U cnt; };
template <class T>
__device__ __forceinline__ T env_tick(Env<T>& e)
{
    using W = V<T>;
    T lv = e.lv + (e.tgt - e.lv) * e.cf;
    const T n = W::u2f(e.cnt), a = -lv, rb = W::rcp(n);
    const T q = a * rb, r = W::fma(-q, n, a), d = W::fma(r, rb, q); // div_near(-lv, n)
    lv = W::fma(e.rs, d, lv);
    e.cnt -= W::usplat(1u);
    e.lv = lv;
    return lv;
}
template <class T>
__device__ __forceinline__ T op_tick(T& phase, T inc, T pm, T env, T level)
{
    using W = V<T>;
    const T o = sinx<T>((phase + pm) * W::splat(6.28318548202514648f)) * env * level;
    const T p = phase + inc;
    phase = p - W::trunc(p);
    return o;
}
template <class T>
__device__ __forceinline__ T voice(const float* in, int stride, int frames, T (*ld)(const float*, int))
{
    using W = V<T>;
    Env<T> e[4];
    T ph[3], inc[3];
    for (int k = 0; k < 4; ++k) {
        e[k].lv = ld(in, 0) * W::splat(0.5f + 0.1f * k);
        e[k].tgt = W::splat(0.0f);
        e[k].cf = W::splat(0.0f);
        e[k].rs = W::splat(1.0f); // releasing: the arithmetic of a realistic score (some lane of the wave always is)
        e[k].cnt = W::usplat(1u << 30);
    }
    for (int k = 0; k < 3; ++k) {
        ph[k] = ld(in, 0) * W::splat(0.3f * (k + 1));
        inc[k] = ld(in, 0) * W::splat(0.01f) + W::splat(0.003f * (k + 1));
    }
    T z0 = W::splat(0.f), z1 = W::splat(0.f), acc = W::splat(0.f);
    const T h = W::splat(0.8f), g = W::splat(0.13f), kk = W::splat(1.5f);
    (void)stride;
#pragma unroll 8
    for (int f = 0; f < frames; ++f) {
        const T e3 = env_tick(e[0]), e2 = env_tick(e[1]), e1 = env_tick(e[2]), ef = env_tick(e[3]);
        const T o3 = op_tick<T>(ph[0], inc[0], W::splat(0.f), e3, W::splat(0.5f));
        const T o2 = op_tick<T>(ph[1], inc[1], o3, e2, W::splat(0.5f));
        const T o1 = op_tick<T>(ph[2], inc[2], o2, e1, W::splat(1.0f));
        const T high = (o1 - z0 * kk - z1) * h, hg = high * g, band = hg + z0, bg = band * g, low = bg + z1;
        z0 = hg + band;
        z1 = bg + low;
        acc += low * W::splat(0.3f) + ef * W::splat(1e-9f);
    }
    return acc + z0 + ph[0] + e[0].lv;
}
__device__ float ld1(const float* in, int) { return in[blockIdx.x * 64 + threadIdx.x]; }
__device__ F2 ld2(const float* in, int) { const int v = blockIdx.x * 128 + threadIdx.x; return (F2){in[v], in[v + 64]}; }
__global__ __launch_bounds__(64) void k1(float* out, const float* in, int frames) { out[blockIdx.x * 64 + threadIdx.x] = voice<float>(in, 0, frames, ld1); }
__global__ __launch_bounds__(64) void k2(float* out, const float* in, int frames)
{
    const F2 r = voice<F2>(in, 0, frames, ld2);
    const int v = blockIdx.x * 128 + threadIdx.x;
    out[v] = r.x;
    out[v + 64] = r.y;
}
int main()
{
    const int frames = 2048;
    hipFuncAttributes a1, a2;
    hipFuncGetAttributes(&a1, (const void*)k1); hipFuncGetAttributes(&a2, (const void*)k2);
    printf("registers: scalar %d, packed %d\n", a1.numRegs, a2.numRegs);
    for (int V : {65536, 131072, 262144, 524288, 1048576}) {
        float *in, *o1, *o2;
        hipMalloc(&in, V * 4); hipMalloc(&o1, V * 4); hipMalloc(&o2, V * 4);
        std::vector<float> h(V);
        for (int i = 0; i < V; ++i) h[i] = 0.1f + (i % 997) / 1100.0f;
        hipMemcpy(in, h.data(), V * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms1 = 0, ms2 = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a); hipLaunchKernelGGL(k1, dim3(V / 64), dim3(64), 0, 0, o1, in, frames); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms1, a, b);
            hipEventRecord(a); hipLaunchKernelGGL(k2, dim3(V / 128), dim3(64), 0, 0, o2, in, frames); hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms2, a, b);
        }
        std::vector<float> r1(V), r2(V);
        hipMemcpy(r1.data(), o1, V * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, V * 4, hipMemcpyDeviceToHost);
        int diff = 0;
        for (int i = 0; i < V; ++i) diff += r1[i] != r2[i];
        printf("%8d voices: scalar %.3f ms (%.3g voices*samples/s), packed %.3f ms (%.3g) -> %.2fx, %d results differ\n", V, ms1,
               (double)V * frames / (ms1 * 1e-3), ms2, (double)V * frames / (ms2 * 1e-3), ms1 / ms2, diff);
        hipFree(in); hipFree(o1); hipFree(o2);
    }
    return 0;
}
