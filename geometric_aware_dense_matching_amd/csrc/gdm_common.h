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

#ifdef __HIPCC__
// Split-bf16 operands (the matching, convolution, up-convolution and circle-loss kernels): v = hi + lo + O(2^-17 |v|), both parts
// rounded to nearest even by the hardware conversion (v_cvt_pk_bf16_f32): a NaN stays a NaN in hi (so it reaches the output, as it
// would through an fp32 product) and an overflow rounds to infinity.  hi / lo come back packed two to a dword (element 0 low).
typedef __attribute__((ext_vector_type(2))) float gdm_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gdm_bf16x2;
__device__ __forceinline__ unsigned gdm_bf16_pk(float a, float b)
{
    const gdm_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gdm_bf16x2));
}
__device__ __forceinline__ void gdm_split2(float a, float b, unsigned& hi, unsigned& lo)
{
    hi = gdm_bf16_pk(a, b);
    lo = gdm_bf16_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ unsigned short gdm_bf16_1(float a) { return (unsigned short)(gdm_bf16_pk(a, 0.f) & 0xffffu); }
#endif
