// og_abi.h -- what every translation unit that defines `extern "C"` entry points of include/oscen_gpu.h shares:
// the thread-local last-error slot, exception types that CARRY their error code (no message sniffing), and the
// guard every entry point that can allocate, compile or touch the device runs under -- nothing may unwind across
// the C ABI ("never throws across the ABI", include/oscen_gpu.h:13).
#pragma once
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

} // namespace ogabi
