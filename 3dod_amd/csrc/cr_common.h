// Shared host-side helpers for libcr3dod.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cr3dod.h"

struct cr_ctx {
    int device;
    hipStream_t stream;
    void* ws;            // small device workspace (flags / counters / split-K slabs)
    size_t ws_bytes;
};

void cr_set_error(const char* fmt, ...);

#define CR_CHECK_ARG(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            cr_set_error(__VA_ARGS__);           \
            return CR_EINVAL;                    \
        }                                        \
    } while (0)

#define CR_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            cr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                         __FILE__, __LINE__);                                     \
            return CR_EHIP;                                                       \
        }                                                                         \
    } while (0)

#define CR_LAUNCH_CHECK() CR_HIP(hipGetLastError())

static inline int64_t cr_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
