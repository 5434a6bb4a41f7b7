// mx_common.h -- shared host-side plumbing for libmemex_hip.so (error slot, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
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

}  // namespace mx
