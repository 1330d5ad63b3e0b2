// Shared helpers for libgdm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/gdm.h"

void gdm_set_error(const char* fmt, ...);

#define GDM_CHECK_ARG(cond, ...)                   \
    do {                                           \
        if (!(cond)) {                             \
            gdm_set_error(__VA_ARGS__);            \
            return GDM_EINVAL;                     \
        }                                          \
    } while (0)

#define GDM_HIP(call)                                                              \
    do {                                                                           \
        hipError_t _e = (call);                                                    \
        if (_e != hipSuccess) {                                                    \
            gdm_set_error("%s failed: %s", #call, hipGetErrorString(_e));          \
            return (int)_e;                                                        \
        }                                                                          \
    } while (0)

static inline int gdm_launch_status(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gdm_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline int gdm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
