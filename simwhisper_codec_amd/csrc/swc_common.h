// Shared helpers for the libswc_hip.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "swc.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;  // raw bf16 storage

void swc_set_error(const char* fmt, ...);

#define SWC_CHECK_ARG(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            swc_set_error(__VA_ARGS__); \
            return SWC_E_ARG;           \
        }                               \
    } while (0)

#define SWC_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            swc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return SWC_E_LAUNCH;                                                 \
        }                                                                        \
    } while (0)

__device__ __forceinline__ bf16_t f32_to_bf16(float x) {
    // plain cast: lowers to v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    __hip_bfloat16 h = __float2bfloat16(x);
    return *reinterpret_cast<bf16_t*>(&h);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t x) {
    return __uint_as_float(((unsigned)x) << 16);
}

template <typename T>
__device__ __forceinline__ void store_out(T* p, float v);
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for bf16 outputs: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution),
// two transcendental ops and ~12 VALU instead of the branchy libm erff.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __expf(-z * z);
    const float erf_abs = 1.0f - poly * t * e;
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
