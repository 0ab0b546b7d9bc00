// og_jit.h -- hiprtc path for graphs that were not compiled ahead of time.
// The same generated translation unit that csrc/gen/ holds for the built-in
// graphs is compiled at og_create() time for gfx950 and loaded as a module.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>

#include "og_graph.h"
#include "og_kernel_rt.hip.h"

struct OgJitKernel {
    virtual ~OgJitKernel() {}
    virtual void launch(const OgBlockArgs& args, bool ramps, bool taps, hipStream_t stream) = 0;
    virtual int occupancy(int depth) = 0; // resident workgroups per CU of the depth-1 / 2 / 4 shape (0: no such shape)
};

// Throws std::runtime_error (compile log included) on failure.
std::unique_ptr<OgJitKernel> og_jit_compile(const ogc::CompiledGraph& cg);
// Compile only (no device needed): returns the code object size; used by the CPU test-suite.
size_t og_jit_compile_only(const ogc::CompiledGraph& cg, const char* arch);
