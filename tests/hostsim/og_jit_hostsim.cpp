// tests/hostsim/og_jit_hostsim.cpp -- TEST INFRASTRUCTURE: og_jit.h for the host simulator.  The library compiles graphs it
// has no ahead-of-time kernel for with hiprtc and loads the code object as a module; here the same generated translation
// unit is compiled for x86 against tests/hostsim/hip/hip_runtime.h into a shared object, loaded with dlopen, and pointed
// at the library's fibre scheduler (simt::Api).  Same kernel names, same grid arithmetic as csrc/og_jit.cpp.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>

#include "og_jit.h"

#ifndef OG_HOSTSIM_JIT_FLAGS
#define OG_HOSTSIM_JIT_FLAGS ""
#endif
#ifndef OG_HOSTSIM_BUILD
#define OG_HOSTSIM_BUILD "_build"
#endif
#ifndef OG_HOSTSIM_DIR
#error "build with -DOG_HOSTSIM_DIR=... -DOG_CSRC_DIR=... -DOG_HOSTSIM_CXX=... (tests/hostsim/build_hostsim.py)"
#endif

namespace {

typedef void (*KernelFn)(OgBlockArgs);

std::string hash_hex(uint64_t h)
{
    char hs[32];
    snprintf(hs, sizeof hs, "%016llx", (unsigned long long)h);
    return hs;
}

// compiles the unit (once per kernel hash: the objects are kept under _build/jit) and returns the path of the shared object
std::string compile_unit(const ogc::CompiledGraph& cg)
{
    const std::string dir = std::string(OG_HOSTSIM_DIR) + "/" OG_HOSTSIM_BUILD "/jit";
    (void)mkdir(dir.c_str(), 0777);
    const std::string base = dir + "/" + hash_hex(cg.hash);
    const std::string so = base + ".so";
    if (std::ifstream(so).good()) return so;
    const std::string src = base + "." + std::to_string((long)getpid()) + ".cpp", tmp = base + "." + std::to_string((long)getpid()) + ".so",
                      log = base + "." + std::to_string((long)getpid()) + ".log";
    {
        std::ofstream f(src);
        f << cg.source << "\n// (host simulator) the loader hands this object the library's scheduler\n"
          << "extern \"C\" void og_hostsim_set_api(void* a) { simt::api_slot() = (simt::Api*)a; }\n";
    }
    const bool debug = getenv("OG_HOSTSIM_DEBUG") != nullptr; // keep the unit and compile it with line tables
    const std::string cmd = std::string(OG_HOSTSIM_CXX) + (debug ? " -g" : "") + OG_HOSTSIM_JIT_FLAGS + " -x c++ -shared -O1 -std=c++17 -ffp-contract=off -fPIC -mfma -mavx2 -DOG_HOSTSIM=1 -DOG_JIT=1"
                            " -Wno-unknown-attributes -Wno-unused-value -Wno-pass-failed -Wno-unknown-pragmas -I" OG_HOSTSIM_DIR " -I" OG_CSRC_DIR " -o " + tmp + " " + src +
                            " > " + log + " 2>&1";
    const int rc = system(cmd.c_str());
    if (rc != 0) {
        std::string text;
        {
            std::ifstream f(log);
            std::string line;
            for (int i = 0; i < 60 && std::getline(f, line); ++i) text += line + "\n";
        }
        (void)remove(tmp.c_str());
        throw std::runtime_error("hostsim jit: compile failed for graph '" + cg.name + "':\n" + text);
    }
    (void)remove(log.c_str());
    if (!debug) (void)remove(src.c_str());
    if (rename(tmp.c_str(), so.c_str()) != 0) throw std::runtime_error("hostsim jit: cannot move " + tmp);
    return so;
}

struct HostJit : OgJitKernel {
    void* dl = nullptr;
    KernelFn fn[4] = {}, fn2[4] = {}, fn4[4] = {}, fn4w[4] = {};
    unsigned lpv = 1;
    ~HostJit() override
    {
        if (dl) dlclose(dl);
    }
    void launch(const OgBlockArgs& args, bool ramps, bool taps, hipStream_t) override
    {
        OgBlockArgs a = args;
        const int vi = (ramps ? 1 : 0) + (taps ? 2 : 0);
        unsigned K = 1;
        if (a.split == 4 && fn4[vi]) K = 4;
        else if (a.split >= 2 && fn2[vi]) K = 2;
        a.split = K > 1 ? K : 0;
        const unsigned grid = K > 1 ? (a.n_voices + OG_WAVE - 1) / OG_WAVE : (unsigned)(((size_t)a.n_voices * lpv + a.lanes - 1) / a.lanes);
        if (!(K == 4 && a.wide && fn4w[vi])) a.wide = 0;
        const KernelFn f = K == 4 ? (a.wide ? fn4w[vi] : fn4[vi]) : (K == 2 ? fn2[vi] : fn[vi]);
        simt::launch(dim3(grid), dim3(K * OG_WAVE), [&]() { f(a); });
    }
    // (what the round-5 fm kernels report on gfx950: registers / LDS of the three shapes)
    int occupancy(int depth) override { return depth == 5 ? (fn4w[0] ? 4 : 0) : (depth == 4 ? (fn4[0] ? 6 : 0) : (depth == 2 ? (fn2[0] ? 8 : 0) : 16)); }
};

} // namespace

size_t og_jit_compile_only(const ogc::CompiledGraph& cg, const char*)
{
    struct stat st;
    const std::string so = compile_unit(cg);
    return stat(so.c_str(), &st) == 0 ? (size_t)st.st_size : 0;
}

std::unique_ptr<OgJitKernel> og_jit_compile(const ogc::CompiledGraph& cg)
{
    const std::string so = compile_unit(cg);
    std::unique_ptr<HostJit> k(new HostJit);
    k->lpv = (unsigned)cg.lpv;
    k->dl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!k->dl) throw std::runtime_error(std::string("hostsim jit: dlopen failed: ") + dlerror());
    auto set_api = (void (*)(void*))dlsym(k->dl, "og_hostsim_set_api");
    if (!set_api) throw std::runtime_error("hostsim jit: og_hostsim_set_api missing");
    set_api((void*)simt::default_api());
    const std::string hs = hash_hex(cg.hash);
    const char* var[4] = {"00", "10", "01", "11"};
    auto sym = [&](const std::string& name) {
        void* p = dlsym(k->dl, name.c_str());
        if (!p) throw std::runtime_error("hostsim jit: kernel " + name + " not found");
        return (KernelFn)p;
    };
    for (int i = 0; i < 4; ++i) {
        k->fn[i] = sym("og_k_" + hs + "_" + var[i]);
        if (cg.max_pipeline >= 2) k->fn2[i] = sym("og_k2_" + hs + "_" + var[i]);
        if (cg.max_pipeline >= 4) k->fn4[i] = sym("og_k4_" + hs + "_" + var[i]);
        if (cg.max_pipeline >= 4 && cg.wide4) k->fn4w[i] = sym("og_k4w_" + hs + "_" + var[i]);
    }
    return k;
}
