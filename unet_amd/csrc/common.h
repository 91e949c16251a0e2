// Shared host/device helpers for libunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "unet_hip.h"

namespace unet {

void set_error(const char* fmt, ...);

#define UNET_CHECK_ARG(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            unet::set_error(__VA_ARGS__);        \
            return UNET_E_BADARG;                \
        }                                        \
    } while (0)

#define UNET_CHECK_HIP(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            unet::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return UNET_E_HIP;                                                           \
        }                                                                                \
    } while (0)

#define UNET_CHECK_LAUNCH()                                                              \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            unet::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return UNET_E_HIP;                                                           \
        }                                                                                \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool slice_ok(int cs, int co, int C) { return cs > 0 && co >= 0 && (cs & 3) == 0 && (co & 3) == 0 && co + C <= cs; }
// v = channels per 16-byte access of the tensor's storage type (4 fp32, 8 bf16)
static inline bool slice_ok_v(int cs, int co, int C, int v) { return cs > 0 && co >= 0 && cs % v == 0 && co % v == 0 && co + C <= cs; }

// hipFuncSetAttribute is per device: one bit per device ordinal in a per-call-site mask (one process may drive several GPUs).
// Returns true the first time the call site runs on the current device.
static inline bool first_use_on_device(unsigned long long* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    const unsigned long long bit = 1ull << (dev & 63);
    return (__atomic_fetch_or(mask, bit, __ATOMIC_RELAXED) & bit) == 0;
}

// grid size for grid-stride HBM-bound kernels (256 CUs x 8 blocks)
static inline int ew_grid(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace unet

namespace unetconv {
// the switches of a launch: the descriptor's struct or the defaults (read-only after load; env overrides: conv_igemm.hip)
const unet_tuning& tuning_defaults();
static inline unet_tuning tuning_of(const unet_tuning* t) { return t != nullptr ? *t : tuning_defaults(); }
}  // namespace unetconv
