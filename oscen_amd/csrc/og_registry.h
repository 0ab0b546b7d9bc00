// og_registry.h -- registry of ahead-of-time compiled voice kernels.
// Every generated translation unit (csrc/gen/*.hip) registers its launch
// function under the FNV-1a hash of its kernel body; og_create() compiles the
// caller's graph description on the host, hashes the body it would generate
// and looks the kernel up here (falling back to hiprtc for unknown graphs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "og_kernel_rt.hip.h"

typedef void (*OgLaunchFn)(const OgBlockArgs& args, bool ramps, bool taps, hipStream_t stream);
// workgroups of the depth-1 / 2 / 4 shape of the kernel a CU can hold at once (registers, LDS): what the engine's choice of
// pipeline depth needs to know (hipOccupancyMaxActiveBlocksPerMultiprocessor); 0 = no such shape
typedef int (*OgOccupancyFn)(int depth);

struct OgKernelEntry {
    uint64_t hash;
    const char* name;
    OgLaunchFn launch;
    OgOccupancyFn occupancy;
    OgKernelEntry* next;
};

OgKernelEntry*& og_kernel_registry_head();
OgLaunchFn og_find_kernel(uint64_t hash);
OgOccupancyFn og_find_occupancy(uint64_t hash);

struct OgKernelRegistrar {
    OgKernelEntry entry;
    OgKernelRegistrar(uint64_t hash, const char* name, OgLaunchFn fn, OgOccupancyFn occ = nullptr)
    {
        entry.hash = hash;
        entry.name = name;
        entry.launch = fn;
        entry.occupancy = occ;
        entry.next = og_kernel_registry_head();
        og_kernel_registry_head() = &entry;
    }
};
