// Shared host-side helpers for libbde2vid.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/bde2vid.h"   // status codes

namespace bde {

std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define BDE_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::bde::fail(BDE_ERR_HIP, "%s failed: %s (%s:%d)", #expr,             \
                               hipGetErrorString(_e), __FILE__, __LINE__);                     \
    } while (0)

#define BDE_TRY(expr)                                                                          \
    do {                                                                                       \
        int _s = (expr);                                                                       \
        if (_s != BDE_OK) return _s;                                                    \
    } while (0)

#define BDE_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) return ::bde::fail(BDE_ERR_ARG, __VA_ARGS__);                      \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

}  // namespace bde
