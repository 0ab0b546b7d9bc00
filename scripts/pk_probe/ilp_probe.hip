// Probe: does a wave with TWO independent voice chains in one instruction stream (scalar f32, not packed)
// hide the dependent-instruction latency that a single chain exposes at one wave per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float sinx(float x)
{
    const float k = __builtin_fmaf(x, 0x1.45f306p-2f, 12582912.0f);
    const float n = k - 12582912.0f;
    float r = __builtin_fmaf(n, -0x1.921fb6p+1f, x);
    r = __builtin_fmaf(n, 0x1.777a5cp-24f, r);
    r = __uint_as_float(__float_as_uint(r) ^ (__float_as_uint(k) << 31));
    const float s = r * r;
    float p = __builtin_fmaf(s, -0x1.9d0bc6p-26f, 0x1.6fadb6p-19f);
    p = __builtin_fmaf(s, p, -0x1.a018e8p-13f);
    p = __builtin_fmaf(s, p, 0x1.111110p-7f);
    p = __builtin_fmaf(s, p, -0x1.555556p-3f);
    return __builtin_fmaf(s * r, p, r);
}
struct Voice {
    float phase, p2, prev, inc, acc;
    __device__ __forceinline__ void tick()
    {
        const float o = sinx((phase + prev * 0.3f) * 6.28318548202514648f) * 0.7f;
        prev = o;
        const float o2 = sinx((p2 + o) * 6.28318548202514648f) * 0.9f;
        float p = phase + inc;
        phase = p - __builtin_truncf(p);
        p = p2 + inc;
        p2 = p - __builtin_truncf(p);
        acc += o2;
    }
};
template <int NV>
__global__ __launch_bounds__(64) void k(float* out, const float* in, int frames)
{
    Voice v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int id = (blockIdx.x * NV + i) * 64 + threadIdx.x;
        v[i] = {in[id], in[id] * 0.5f, 0.f, in[id] * 0.01f + 0.003f, 0.f};
    }
    for (int f = 0; f < frames; ++f) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i].tick();
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) out[(blockIdx.x * NV + i) * 64 + threadIdx.x] = v[i].acc + v[i].prev;
}
// The same two chains, interleaved by hand statement by statement (what a latency-aware scheduler would do).
#define SIN_STEPS(T)                                                      \
    T(k, = __builtin_fmaf(x, 0x1.45f306p-2f, 12582912.0f))                \
    T(n, = k - 12582912.0f)                                               \
    T(r, = __builtin_fmaf(n, -0x1.921fb6p+1f, x))                         \
    T(r, = __builtin_fmaf(n, 0x1.777a5cp-24f, r))                         \
    T(r, = __uint_as_float(__float_as_uint(r) ^ (__float_as_uint(k) << 31))) \
    T(s, = r * r)                                                         \
    T(p, = __builtin_fmaf(s, -0x1.9d0bc6p-26f, 0x1.6fadb6p-19f))          \
    T(p, = __builtin_fmaf(s, p, -0x1.a018e8p-13f))                        \
    T(p, = __builtin_fmaf(s, p, 0x1.111110p-7f))                          \
    T(p, = __builtin_fmaf(s, p, -0x1.555556p-3f))                         \
    T(y, = __builtin_fmaf(s * r, p, r))
__global__ __launch_bounds__(64) void k2i(float* out, const float* in, int frames)
{
    const int ia = (blockIdx.x * 2) * 64 + threadIdx.x, ib = ia + 64;
    float pha = in[ia], phb = in[ib], p2a = pha * 0.5f, p2b = phb * 0.5f, pva = 0.f, pvb = 0.f;
    const float inca = in[ia] * 0.01f + 0.003f, incb = in[ib] * 0.01f + 0.003f;
    float acca = 0.f, accb = 0.f;
    for (int f = 0; f < frames; ++f) {
        float xa = (pha + pva * 0.3f) * 6.28318548202514648f, xb = (phb + pvb * 0.3f) * 6.28318548202514648f;
        float ka, na, ra, sa, pa, ya, kb, nb, rb, sb, pb, yb;
#define T(v, e) { const float x = xa, k = ka, n = na, r = ra, s = sa, p = pa; (void)x; (void)k; (void)n; (void)r; (void)s; (void)p; v##a e; } \
                { const float x = xb, k = kb, n = nb, r = rb, s = sb, p = pb; (void)x; (void)k; (void)n; (void)r; (void)s; (void)p; v##b e; }
        SIN_STEPS(T)
        const float oa = ya * 0.7f, ob = yb * 0.7f;
        pva = oa; pvb = ob;
        xa = (p2a + oa) * 6.28318548202514648f; xb = (p2b + ob) * 6.28318548202514648f;
        SIN_STEPS(T)
#undef T
        const float o2a = ya * 0.9f, o2b = yb * 0.9f;
        float ta = pha + inca, tb = phb + incb;
        pha = ta - __builtin_truncf(ta); phb = tb - __builtin_truncf(tb);
        ta = p2a + inca; tb = p2b + incb;
        p2a = ta - __builtin_truncf(ta); p2b = tb - __builtin_truncf(tb);
        acca += o2a; accb += o2b;
    }
    out[ia] = acca + pva;
    out[ib] = accb + pvb;
}

int main()
{
    const int frames = 4096, V = 65536;
    float *in, *o;
    hipMalloc(&in, V * 4); hipMalloc(&o, V * 4);
    std::vector<float> h(V);
    for (int i = 0; i < V; ++i) h[i] = (i % 997) / 997.0f;
    hipMemcpy(in, h.data(), V * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[3] = {0, 0, 0};
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL(k<1>, dim3(V / 64), dim3(64), 0, 0, o, in, frames); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[0], a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k<2>, dim3(V / 128), dim3(64), 0, 0, o, in, frames); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[1], a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k<4>, dim3(V / 256), dim3(64), 0, 0, o, in, frames); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[2], a, b);
    }
    float msi = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL(k2i, dim3(V / 128), dim3(64), 0, 0, o, in, frames); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&msi, a, b);
    }
    printf("2 chains/wave interleaved by hand (512 waves) %.3f ms\n", msi);
    printf("65536 voices, 4096 frames: 1 chain/wave (1024 waves) %.3f ms | 2 chains/wave (512 waves) %.3f ms | 4 chains/wave (256 waves) %.3f ms\n", ms[0], ms[1], ms[2]);
    return 0;
}
