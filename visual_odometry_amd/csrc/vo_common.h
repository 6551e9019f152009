// vo_common.h -- shared by the translation units of libvo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vo_hip.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

void vo_set_error(const char* fmt, ...);

#define VO_CHECK_ARG(cond, code, ...)                 \
    do {                                              \
        if (!(cond)) {                                \
            vo_set_error(__VA_ARGS__);                \
            return (code);                            \
        }                                             \
    } while (0)

#define VO_HIP_TRY(expr)                                                              \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            vo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                         __FILE__, __LINE__);                                         \
            return VO_EHIP;                                                           \
        }                                                                             \
    } while (0)

#define VO_LAUNCH_CHECK() VO_HIP_TRY(hipGetLastError())

static inline int vo_ceil_div(int a, int b) { return (a + b - 1) / b; }
