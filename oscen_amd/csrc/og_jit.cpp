// og_jit.cpp -- see og_jit.h
#include "og_jit.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

bool readable(const std::string& path) { return std::ifstream(path).good(); }

// the kernel headers ship next to the library: <libdir>/csrc (or one level up for build variants)
std::string csrc_dir()
{
    if (const char* e = getenv("OSCEN_GPU_CSRC")) return e;
    Dl_info info;
    if (dladdr((void*)&csrc_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        const std::string dir = (s == std::string::npos ? std::string(".") : p.substr(0, s));
        for (const char* rel : {"/csrc", "/../csrc"})
            if (readable(dir + rel + "/og_kernel_rt.hip.h")) return dir + rel;
    }
    return "csrc";
}

std::string slurp(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("oscen jit: cannot read " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::vector<char> compile_to_code(const ogc::CompiledGraph& cg, const char* arch)
{
    const std::string dir = csrc_dir();
    const char* names[3] = {"og_kernel_rt.hip.h", "og_nodes.hip.h", "og_math.h"};
    std::string bodies[3];
    const char* srcs[3];
    for (int i = 0; i < 3; ++i) {
        bodies[i] = slurp(dir + "/" + names[i]);
        srcs[i] = bodies[i].c_str();
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, cg.source.c_str(), (cg.name + ".hip").c_str(), 3, srcs, names) != HIPRTC_SUCCESS)
        throw std::runtime_error("oscen jit: hiprtcCreateProgram failed");
    std::string archopt = std::string("--offload-arch=") + arch;
    const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DOG_JIT=1"};
    hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        throw std::runtime_error("oscen jit: compile failed for graph '" + cg.name + "':\n" + log);
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    std::vector<char> code(n);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return code;
}

struct JitImpl : OgJitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn[4] = {};
    hipFunction_t fn2[4] = {}; // two-wave pipeline variants (when the graph has them)
    hipFunction_t fn4[4] = {}; // four-wave pipeline variants
    hipFunction_t fn4w[4] = {}; // ... with 16-frame hand-offs (when the graph has them)
    unsigned lpv = 1;
    ~JitImpl() override
    {
        if (mod) (void)hipModuleUnload(mod);
    }
    void launch(const OgBlockArgs& args, bool ramps, bool taps, hipStream_t stream) override
    {
        OgBlockArgs a = args;
        size_t sz = sizeof a;
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        const int vi = (ramps ? 1 : 0) + (taps ? 2 : 0);
        unsigned K = 1;
        if (a.split == 4 && fn4[vi]) K = 4;
        else if (a.split >= 2 && fn2[vi]) K = 2;
        a.split = K > 1 ? K : 0;
        const unsigned grid = K > 1 ? (a.n_voices + OG_WAVE - 1) / OG_WAVE
                                    : (unsigned)(((size_t)a.n_voices * lpv + a.lanes - 1) / a.lanes);
        if (!(K == 4 && a.wide && fn4w[vi])) a.wide = 0;
        hipError_t e = hipModuleLaunchKernel(K == 4 ? (a.wide ? fn4w[vi] : fn4[vi]) : (K == 2 ? fn2[vi] : fn[vi]), grid, 1, 1, K * OG_WAVE, 1, 1, 0,
                                             stream, nullptr, cfg);
        if (e != hipSuccess) throw std::runtime_error(std::string("oscen jit: launch failed: ") + hipGetErrorString(e));
    }
    int occupancy(int depth) override
    {
        hipFunction_t f = depth == 5 ? fn4w[0] : (depth == 4 ? fn4[0] : (depth == 2 ? fn2[0] : fn[0]));
        int n = 0;
        const int waves = depth == 5 ? 4 : (depth >= 2 ? depth : 1);
        if (!f || hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, waves * OG_WAVE, 0) != hipSuccess) return 0;
        return n;
    }
};

} // namespace

size_t og_jit_compile_only(const ogc::CompiledGraph& cg, const char* arch) { return compile_to_code(cg, arch).size(); }

std::unique_ptr<OgJitKernel> og_jit_compile(const ogc::CompiledGraph& cg)
{
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        throw std::runtime_error("oscen jit: no device");
    std::string arch = prop.gcnArchName;
    size_t colon = arch.find(':');
    if (colon != std::string::npos) arch = arch.substr(0, colon);
    std::vector<char> code = compile_to_code(cg, arch.c_str());
    std::unique_ptr<JitImpl> k(new JitImpl);
    k->lpv = (unsigned)cg.lpv;
    if (hipModuleLoadData(&k->mod, code.data()) != hipSuccess) throw std::runtime_error("oscen jit: hipModuleLoadData failed");
    char hs[32];
    snprintf(hs, sizeof hs, "%016llx", (unsigned long long)cg.hash);
    const char* var[4] = {"00", "10", "01", "11"};
    for (int i = 0; i < 4; ++i) {
        std::string name = std::string("og_k_") + hs + "_" + var[i];
        if (hipModuleGetFunction(&k->fn[i], k->mod, name.c_str()) != hipSuccess)
            throw std::runtime_error("oscen jit: kernel " + name + " not found in module");
        if (cg.max_pipeline >= 2) {
            const std::string name2 = std::string("og_k2_") + hs + "_" + var[i];
            if (hipModuleGetFunction(&k->fn2[i], k->mod, name2.c_str()) != hipSuccess)
                throw std::runtime_error("oscen jit: kernel " + name2 + " not found in module");
        }
        if (cg.max_pipeline >= 4) {
            const std::string name4 = std::string("og_k4_") + hs + "_" + var[i];
            if (hipModuleGetFunction(&k->fn4[i], k->mod, name4.c_str()) != hipSuccess)
                throw std::runtime_error("oscen jit: kernel " + name4 + " not found in module");
            if (cg.wide4) {
                const std::string name4w = std::string("og_k4w_") + hs + "_" + var[i];
                if (hipModuleGetFunction(&k->fn4w[i], k->mod, name4w.c_str()) != hipSuccess)
                    throw std::runtime_error("oscen jit: kernel " + name4w + " not found in module");
            }
        }
    }
    return k;
}
