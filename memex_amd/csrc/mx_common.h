// mx_common.h -- shared host-side plumbing for libmemex_hip.so (error slot, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <exception>
#include <new>
#include <string>

#include "../../include/memex_hip.h"

namespace mx {

// thread-local error text behind mx_last_error()
std::string &last_error_slot();

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

// The C ABI never lets a C++ exception cross into the caller (a Rust or ctypes frame cannot unwind it): every `int mx_*`
// entry point is a function-try-block that ends in `catch (...) { return guard_exception(); }`.
inline int guard_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(MX_ENOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(MX_EDEVICE, "internal error: %s", e.what());
    } catch (...) {
        return fail(MX_EDEVICE, "internal error");
    }
}

#define MX_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return ::mx::fail(_e == hipErrorOutOfMemory ? MX_ENOMEM : MX_EDEVICE, "%s: %s (%s:%d)", \
                              #call, hipGetErrorString(_e), __FILE__, __LINE__);                  \
    } while (0)

// RAII: make `dev` current for the scope, restore the previous device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

inline uint64_t round_up(uint64_t x, uint64_t m) { return (x + m - 1) / m * m; }

// Host wait for everything queued on `st` that does not burn a core: hipStreamSynchronize / hipEventSynchronize poll
// with the runtime's default scheduling policy (100 % of a core, scripts/gpu_wait_modes.py).  Record `ev`, poll it
// for `spin_us` (short work returns at once), then nap 50 us between polls.  MEMEX_HIP_SPIN=1: plain synchronize.
inline hipError_t napping_sync(hipStream_t st, hipEvent_t ev, int spin_us = 100) {
    static const bool spin = [] {
        const char *sp = getenv("MEMEX_HIP_SPIN");
        return sp && sp[0] == '1';
    }();
    if (spin || !ev) return hipStreamSynchronize(st);
    hipError_t e = hipEventRecord(ev, st);
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) {
            struct timespec ts = {0, 50 * 1000};
            (void)clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
        }
    }
}

}  // namespace mx
