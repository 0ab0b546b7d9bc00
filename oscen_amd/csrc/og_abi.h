// og_abi.h -- what every translation unit that defines `extern "C"` entry points of include/oscen_gpu.h shares:
// the thread-local last-error slot, exception types that CARRY their error code (no message sniffing), and the
// guard every entry point that can allocate, compile or touch the device runs under -- nothing may unwind across
// the C ABI ("never throws across the ABI", include/oscen_gpu.h:13).
#pragma once
#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>

#include "../../include/oscen_gpu.h"

namespace ogabi {

// an error with its OG_E_* code: thrown where the failure is diagnosed, mapped 1:1 by guard()
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
// a graph that uses a feature this build lacks (OG_E_UNSUPPORTED): thrown by the graph compiler / DSL front end
struct Unsupported : Error {
    explicit Unsupported(const std::string& m) : Error(OG_E_UNSUPPORTED, m) {}
};
// a HIP / RCCL call failed (OG_E_DEVICE)
struct DeviceError : Error {
    explicit DeviceError(const std::string& m) : Error(OG_E_DEVICE, m) {}
};

// records `msg` as this thread's og_last_error() and returns `code` (defined in og_engine.cpp)
int set_error(int code, const std::string& msg);
int set_error(int code, const char* msg) noexcept; // (no std::string temporary at the call site: what guard() calls)

template <class F>
int guard(F&& f) noexcept
{
    try {
        return f();
    } catch (const Error& e) {
        return set_error(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return set_error(OG_E_NOMEM, "out of host memory");
    } catch (const std::exception& e) { // malformed graph / bad argument diagnosed by the compiler front end
        return set_error(OG_E_INVALID, e.what());
    } catch (...) {
        return set_error(OG_E_INVALID, "unknown exception");
    }
}

// the same for entry points that return a value instead of a code (sizes, counters): `fallback` on failure
template <class T, class F>
T guard_value(T fallback, F&& f) noexcept
{
    const int rc = guard([&]() -> int {
        fallback = f();
        return OG_OK;
    });
    (void)rc;
    return fallback;
}

// ---- environment knobs -----------------------------------------------------------------------------------------------
// SETTINGS are part of the product and read as they are: OSCEN_GPU_SPLIT (pin the kernel shape: 0 ordinary, 2 / 4 waves per
// 64 voices), OSCEN_GPU_WIDE (16-frame hand-offs of the four-wave shape on / off), OSCEN_GPU_RCCL_LIB (path of the librccl to
// bind), OSCEN_GPU_CSRC (where the JIT finds the device headers).  Everything else is an EXPERIMENT knob of the A/B scripts
// (scripts/build_variant.py, scripts/ab_bench.sh) and of a few tests: read through experiment_knob(), which IGNORES the
// environment unless OSCEN_GPU_EXPERIMENTAL=1 -- a stray variable cannot change what a product build generates or runs.
// og_version() lists both sets; tests/test_abi_guard_cpu.py checks that no translation unit reads any other variable.
#define OG_SETTINGS "OSCEN_GPU_SPLIT OSCEN_GPU_WIDE OSCEN_GPU_RCCL_LIB OSCEN_GPU_CSRC OSCEN_GPU_EXPERIMENTAL"
#define OG_EXPERIMENT_KNOBS                                                                                                      \
    "OSCEN_GPU_HOST_PROF OSCEN_GPU_BLOCKING_MEMCPY OSCEN_GPU_EV_HEADROOM OSCEN_GPU_LANES OSCEN_GPU_OUT_EVENTS OSCEN_GPU_FORCE_RCCL "  \
    "OGC_TPT_FLAT OGC_TPT_LAZY OGC_HPL OGC_ALAP OGC_SPLIT OGC_CUT2 OGC_PARTS OGC_K3 OGC_CUTS OGC_UNROLL OGC_PRIO_PARITY OGC_CHUNK_CHK "  \
    "OGC_STICKY1 OGC_XCH16 OGC_XCH OGC_ROT OGC_PRIO OGC_STICKY OGC_FORCE_PATH OGC_EVSKIP OGC_EVUNROLL OGC_RELPRIO OGC_SLOWPRIO OGC_WAVES_EU OGC_FLAGS OGC_NARROW_FD OGC_STAGEEND_BODY"
inline bool experiments_enabled()
{
    const char* e = getenv("OSCEN_GPU_EXPERIMENTAL");
    return e && atoi(e) != 0;
}
inline const char* experiment_knob(const char* name) { return experiments_enabled() ? getenv(name) : nullptr; }

} // namespace ogabi
