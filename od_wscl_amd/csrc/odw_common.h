// odw_common.h -- shared helpers for libodwscl.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/odwscl.h"

#define ODW_EXPORT extern "C" __attribute__((visibility("default")))

// thread-local last-error message (with the per-kernel LDS-size cache below: the only mutable state in the library)
void odw_set_error(const char* fmt, ...);

#define ODW_REQUIRE(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            odw_set_error(__VA_ARGS__);           \
            return ODW_EINVAL;                    \
        }                                         \
    } while (0)

#define ODW_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            odw_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return ODW_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call on the launch path of every kernel that asks for more
// than 64 KB of LDS -- ~26 GEMM launches per step.  Remember the largest size set per (device, kernel) -- the attribute is a
// per-device property of the function -- and only go to the driver for more.  Callers evaluate this on EVERY launch (no
// function-local `static` results: a process that later launches on a second GPU must set the attribute there too).
#include <mutex>
#include <unordered_map>
static inline hipError_t odw_set_max_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, int> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t key = (uint64_t)(uintptr_t)fn ^ ((uint64_t)(unsigned)dev << 56);
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(key);
    if (it != seen.end() && it->second >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) seen[key] = bytes;
    return e;
}

#define ODW_CHECK_HIP(expr, name)                                        \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            odw_set_error("%s: %s", name, hipGetErrorString(e_));        \
            return ODW_ELAUNCH;                                          \
        }                                                                \
    } while (0)

__host__ __device__ static inline int64_t odw_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// MI355X constants used for launch geometry
constexpr int ODW_NUM_CU = 256;
constexpr int ODW_LDS_BYTES = 160 * 1024;
constexpr int ODW_WAVE = 64;
