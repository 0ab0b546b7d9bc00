// tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets the engine's host code AND the generated voice kernels be compiled for
// the x86 host, so that the control logic of the kernels (chunk variants, sticky loops, hand-off barriers of the
// pipelined kernels, event walks, bus tiles) can be executed by `pytest -m "not gpu"` in this GPU-less container --
// through the same C ABI, against the same oracle, as the `-m gpu` parity tests do on the MI355X.
//
//   * a kernel launch runs every lane of every workgroup as a fibre (ucontext); the wave-level operations the kernels
//     use (__all, __any, __shfl_xor, readfirstlane, the wave barrier) and __syncthreads are rendezvous points
//     (tests/hostsim/simt.cpp); a collective reached by only part of a wave is reported as a deadlock, not guessed at;
//   * the runtime API is synchronous: device memory is host memory, a stream executes at enqueue time (issue order is
//     one of the schedules a correct stream program must allow), events are timestamps.
//
// What this is NOT: a CPU fallback.  oscen_amd/ never builds, names or loads it; liboscen_gpu.so has no host path and
// fails loudly without a device.  Only tests/test_hostsim_cpu.py builds tests/hostsim/_build/liboscen_gpu_hostsim.so and
// points OSCEN_GPU_LIB (the A/B hook of the bindings) at it, in a subprocess.  Numbers differ from the GPU's in the last
// bits (libm instead of ocml, 1/x instead of v_rcp_f32); the tests compare against the oracle at the contract's 1e-5.
#pragma once
#ifndef OG_HOSTSIM
#define OG_HOSTSIM 1
#endif

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

// ---- language ---------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static // one workgroup runs at a time (simt::launch holds a process-wide lock)
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...) unused // (`__attribute__((amdgpu_waves_per_eu(n)))` on the generated kernels)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 {
    unsigned x, y;
};
struct uint4 {
    unsigned x, y, z, w;
};
struct float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// HIP's device-side integer min / max
template <class T>
static inline T min(T a, T b)
{
    return b < a ? b : a;
}
template <class T>
static inline T max(T a, T b)
{
    return a < b ? b : a;
}

namespace simt {
struct Idx {
    unsigned x, y, z;
};
struct LaneCtx {
    Idx tid, bid, bdim, gdim;
};
// every shared object built from this header (the library, a kernel compiled at run time) reaches the ONE scheduler of
// the library through this table
struct Api {
    LaneCtx* (*cur)();
    int (*all)(int);
    int (*any)(int);
    uint32_t (*shfl_xor)(uint32_t, int);
    uint32_t (*readfirstlane)(uint32_t);
    void (*wave_sync)();
    void (*syncthreads)();
    void (*launch)(dim3, dim3, const std::function<void()>&);
    void (*yield)(); // a lane that polls shared memory (the flag hand-off of the pipelined kernels) lets the others run
};
Api* default_api(); // simt.cpp (library only)
inline Api*& api_slot()
{
    static Api* a = nullptr;
    return a;
}
inline Api* api()
{
    Api* a = api_slot();
    return a;
}
inline void launch(dim3 g, dim3 b, const std::function<void()>& body) { api()->launch(g, b, body); }
} // namespace simt

#define threadIdx (simt::api()->cur()->tid)
#define blockIdx (simt::api()->cur()->bid)
#define blockDim (simt::api()->cur()->bdim)
#define gridDim (simt::api()->cur()->gdim)

static inline int __all(int p) { return simt::api()->all(p); }
static inline int __any(int p) { return simt::api()->any(p); }
static inline void __syncthreads() { simt::api()->syncthreads(); }
static inline float __shfl_xor(float v, int m)
{
    uint32_t b;
    memcpy(&b, &v, 4);
    b = simt::api()->shfl_xor(b, m);
    memcpy(&v, &b, 4);
    return v;
}
static inline uint32_t __shfl_xor(uint32_t v, int m) { return simt::api()->shfl_xor(v, m); }
static inline int __shfl_xor(int v, int m) { return (int)simt::api()->shfl_xor((uint32_t)v, m); }
static inline float __uint_as_float(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t __float_as_uint(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
template <class T>
static inline T atomicAdd(T* p, T v)
{
    const T old = *p; // (lanes are cooperative fibres of one thread: nothing runs between the read and the write)
    *p = old + v;
    return old;
}
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline void __threadfence_system() {}
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(ptr, val, order, scope) __atomic_store_n((ptr), (val), (order))

// the amdgcn builtins the device headers call by name
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(n) (simt::api()->yield())
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_load(ptr, order, scope) __atomic_load_n((ptr), (order))
#define __builtin_amdgcn_wave_barrier() (simt::api()->wave_sync())
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
// (the kernels call it on wave-uniform values only -- kernel-argument slots, to pin them to a scalar register -- and from
//  divergent code, where only the active lanes take part: the caller's own value IS the first active lane's)
#define __builtin_amdgcn_readfirstlane(x) (x)
static inline float og_hostsim_med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) og_hostsim_med3((a), (b), (c))

// ---- runtime API (synchronous) --------------------------------------------------------------------------------------
typedef int hipError_t;
enum {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorInvalidDevice = 101,
    hipErrorNotReady = 600,
    hipErrorNotSupported = 801,
};
struct ihipStream_t {
    int device;
    unsigned flags;
};
struct ihipEvent_t {
    double t_ms;
    bool recorded;
};
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;
typedef void* hipModule_t;
typedef void* hipFunction_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int clockRate;
    int warpSize;
    size_t sharedMemPerBlock;
    int maxThreadsPerBlock;
};

extern "C" { // (C linkage like the real runtime: a test reaches hipMalloc & co. through the loaded library's handle)
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void* dst, int v, size_t n);
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
// modules (the hiprtc path): kernels compiled at run time are shared objects here, see tests/hostsim/og_jit_hostsim.cpp
hipError_t hipModuleLoadData(hipModule_t* m, const void* image);
hipError_t hipModuleUnload(hipModule_t m);
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name);
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shmem,
                                 hipStream_t s, void** params, void** extra);

} // extern "C"
// occupancy of an ahead-of-time kernel: what the round-5 fm kernels report on gfx950 for 64- / 128- / 256-thread workgroups
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int block, size_t)
{
    *n = block >= 256 ? 6 : (block >= 128 ? 8 : 16); // (the wide four-wave form reports 6 here, 4 on the hardware: results do not depend on it)
    return hipSuccess;
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n)
{
    return hipMalloc((void**)p, n);
}
template <class T>
static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0)
{
    return hipHostMalloc((void**)p, n, flags);
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    simt::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
