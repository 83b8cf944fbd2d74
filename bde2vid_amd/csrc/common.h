// Shared host-side helpers for libbde2vid.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
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

// split.h: the operand format the split kernels take unless bde_set_tuning("sb_terms") says otherwise
#define BDE_DEFAULT_SB_TERMS 2

// Launch-shape overrides of one model (bde_set_tuning).  The launch helpers in the kernel headers read them through
// a thread-local pointer that every C-ABI entry point sets to ITS model for the duration of the call, so two models
// (or two devices) in one process do not see each other's settings.
struct Tuning {
    int attn_mfma = 1;      // 0 = attn.h for every head_dim
    int conv_nt = 0;        // force 32-pixel tiles per wave (1 | 2), 0 = auto
    int conv_vec = 1;       // float4-staged convs
    int lstm_shape = 0;     // seg*100000 + rows*1000 + pxw, 0 = auto
    int pw_force = 0;       // mt*100 + nt*10 + ws, 0 = auto
    int pw_batched = 0;     // same for the T-batched launches
    int tok_npt = 0;        // 0 = auto, 1 | 2 = forced
};
inline const Tuning*& tuning_tls() {
    static const Tuning defaults;
    static thread_local const Tuning* cur = &defaults;
    return cur;
}
inline const Tuning& tuning() { return *tuning_tls(); }
struct TuningScope {
    const Tuning* prev;
    explicit TuningScope(const Tuning* t) : prev(tuning_tls()) { tuning_tls() = t; }
    ~TuningScope() { tuning_tls() = prev; }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: remember per device (not per
// process) that it has been raised.  `seen` is one static array per kernel instantiation.  Safe against concurrent
// forwards from several host threads: the flag is published (release) only after the attribute is set, and the slow
// path is serialised.
constexpr int BDE_MAX_DEVICES = 64;
inline std::mutex& lds_attr_mutex() {
    static std::mutex mu;
    return mu;
}
inline hipError_t raise_dynamic_lds(unsigned char (&seen)[BDE_MAX_DEVICES], const void* kernel, int bytes = 160 * 1024) {
    int d = 0;
    const bool indexed = hipGetDevice(&d) == hipSuccess && d >= 0 && d < BDE_MAX_DEVICES;
    if (indexed && __atomic_load_n(&seen[d], __ATOMIC_ACQUIRE)) return hipSuccess;
    std::lock_guard<std::mutex> lock(lds_attr_mutex());
    if (indexed && __atomic_load_n(&seen[d], __ATOMIC_ACQUIRE)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && indexed) __atomic_store_n(&seen[d], (unsigned char)1, __ATOMIC_RELEASE);
    return e;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

}  // namespace bde
