// Shared helpers for the libswc_hip.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "swc.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;  // raw bf16 storage
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct f16s_t { unsigned short h; };  // tag type: split-f16 storage (see SWC_F16S in swc.h), addressed in halves
struct fp8_t { unsigned char b; };    // OCP e4m3fn byte (SWC_FP8)

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
// two f32 -> one dword (lo in bits 0..15, hi in 16..31), RNE: ONE v_cvt_pk_bf16_f32 (the `cast | cast << 16` form compiles
// to one conversion per value plus the merge).  A vector conversion, not inline asm: hipcc then still inserts the wait
// state a transcendental result needs before its first use.
typedef float f32x2_c __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack2(float lo, float hi) {
    const bf16x2_c r = __builtin_convertvector((f32x2_c){lo, hi}, bf16x2_c);
    return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t x) {
    return __uint_as_float(((unsigned)x) << 16);
}

// Saturation accounting (swc_set_saturation_counter in swc.h).  Producers of split-f16 / fp8 ACTIVATIONS keep the
// largest |scaled value| they converted in a register (one v_max per element, |x| is a free source modifier) and add
// one to the caller's device counter when it exceeded the format's range: the conversion then clipped.
#define SWC_F16S_LIMIT 65504.0f
#define SWC_FP8_LIMIT 448.0f
unsigned* swc_sat_counter();  // host: the calling thread's counter pair {split-f16, fp8} (device memory) or nullptr
__device__ __forceinline__ void sat_commit(unsigned* sat, int which, float amax, float limit) {
    if (sat != nullptr && amax > limit) atomicAdd(sat + which, 1u);
}

// split one scaled f32 into (hi, lo) halves; saturates instead of overflowing to inf
__device__ __forceinline__ void f16s_split(float v, unsigned short& hi, unsigned short& lo) {
    v = fminf(fmaxf(v, -SWC_F16S_LIMIT), SWC_F16S_LIMIT);
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    hi = *reinterpret_cast<const unsigned short*>(&h);
    lo = *reinterpret_cast<const unsigned short*>(&l);
}
// half index of logical column k inside a split row (block of 32: 32 hi then 32 lo)
__device__ __forceinline__ long f16s_col(long k) { return (k >> 5) * 64 + (k & 31); }
// two values at once -> one dword of hi halves, one of lo halves: the conversions are the packed forms (one
// v_cvt_pk_f16_f32 per pair, RNE like the scalar casts), the clamp one v_med3 each, the range tracking one v_max3 for both.
// Bit-identical to two f16s_split calls.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16s_split2(float a, float b, unsigned& hi2, unsigned& lo2, float& amax) {
    amax = fmaxf(fmaxf(amax, fabsf(a)), fabsf(b));
    a = __builtin_amdgcn_fmed3f(a, -SWC_F16S_LIMIT, SWC_F16S_LIMIT);
    b = __builtin_amdgcn_fmed3f(b, -SWC_F16S_LIMIT, SWC_F16S_LIMIT);
    const f16x2_t h = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
    const f16x2_t l = __builtin_convertvector((f32x2_t){a - (float)h[0], b - (float)h[1]}, f16x2_t);
    hi2 = *reinterpret_cast<const unsigned*>(&h);
    lo2 = *reinterpret_cast<const unsigned*>(&l);
}

// store 4 consecutive logical columns k..k+3 (k % 4 == 0) of a split row starting at `row` (halves)
__device__ __forceinline__ void f16s_store4(unsigned short* row, long k, float a, float b, float c, float d, float& amax) {
    unsigned h0, l0, h1, l1;
    f16s_split2(a, b, h0, l0, amax);
    f16s_split2(c, d, h1, l1, amax);
    unsigned short* p = row + f16s_col(k);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + 32) = make_uint2(l0, l1);
}
__device__ __forceinline__ void f16s_store4(unsigned short* row, long k, float a, float b, float c, float d) {
    float unused = 0.f;
    f16s_store4(row, k, a, b, c, d, unused);
}

// the same with range tracking
__device__ __forceinline__ void f16s_split(float v, unsigned short& hi, unsigned short& lo, float& amax) {
    amax = fmaxf(amax, fabsf(v));
    f16s_split(v, hi, lo);
}

// 4 floats -> 4 e4m3 bytes (v_cvt_pk_fp8_f32, RNE), saturating at the largest finite value
__device__ __forceinline__ unsigned fp8_pack4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

__device__ __forceinline__ unsigned fp8_pack4(float a, float b, float c, float d, float& amax) {
    amax = fmaxf(fmaxf(amax, fabsf(a)), fmaxf(fmaxf(fabsf(b), fabsf(c)), fabsf(d)));
    return fp8_pack4(a, b, c, d);
}

template <typename T>
__device__ __forceinline__ void store_out(T* p, float v);
template <>
__device__ __forceinline__ void store_out<fp8_t>(fp8_t* p, float v) { p->b = (unsigned char)(fp8_pack4(v, 0.f, 0.f, 0.f) & 0xff); }
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for bf16 OUTPUTS only: x * sigmoid(2u), u = 0.80015708 (x + 0.0433676 x^3) — the tanh form with its two
// constants re-fitted (minimax) to the exact erf GELU: |error| <= 2.7e-4 everywhere, against a bf16 output
// rounding of up to 7.8e-3 on |x| < 4.  5 VALU + 2 transcendental ops (36 issue cycles; a transcendental-free 7-term
// packed polynomial, |error| 1.5e-4, was measured slower: v_pk_*_f32 issues at half rate); the exact-erf path costs ~4x that and made
// the pwconv1 / fc1 epilogues VALU-bound.  f32 / split-f16 outputs keep erff.
__device__ __forceinline__ float gelu_fast(float x) {
    const float k0 = 0.80015708f * 2.885390081777927f;              // 2 log2(e) * c0
    const float k1 = 0.80015708f * 0.0433676f * 2.885390081777927f;
    const float e = __builtin_amdgcn_exp2f(x * fmaf(k1, x * x, k0));  // exp(2u); inf / 0 at the extremes are fine
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    return fmaf(-x, r, x);
}

// GELU for split-f16 outputs (f32-class path): erf by Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7 (the size of
// an f32 rounding of a value near 1), branch-free, ~17 VALU + 2 transcendental ops instead of libm's erff.
__device__ __forceinline__ float gelu_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    // t = 1 / (1 + p z): hardware reciprocal + one Newton step (<= 1 ulp; the IEEE-rounded division costs 12
    // instructions per output and bought nothing against the formula's own 1.5e-7)
    const float d = fmaf(0.3275911f, z, 1.0f);
    float t = __builtin_amdgcn_rcpf(d);
    t = fmaf(t, fmaf(-d, t, 1.0f), t);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-z^2) = 2^(-x^2 log2(e) / 2)
    const float erf_abs = fmaf(-poly * t, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// One-time (per kernel instantiation and per device) opt-in to more than 64 KiB of dynamic LDS.  `done` is a
// per-call-site static array indexed by device; a race only repeats the idempotent attribute call.
#define SWC_ENABLE_LDS(kern, bytes, who)                                                                    \
    do {                                                                                                    \
        static bool done__[64] = {};                                                                        \
        int dev__ = 0;                                                                                      \
        (void)hipGetDevice(&dev__);                                                                         \
        const int slot__ = (dev__ >= 0 && dev__ < 64) ? dev__ : 0;                                          \
        if (!done__[slot__] || dev__ >= 64) {                                                               \
            hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                       \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));      \
            if (e__ != hipSuccess) {                                                                        \
                swc_set_error("%s: cannot enable %d bytes of LDS: %s", who, (int)(bytes), hipGetErrorString(e__)); \
                return SWC_E_LAUNCH;                                                                        \
            }                                                                                               \
            done__[slot__] = true;                                                                          \
        }                                                                                                   \
    } while (0)

// Sum over the 64 lanes on the DPP path (no LDS crossbar round trips): xor 1, xor 2, half-row mirror and row
// mirror leave each row of 16 lanes holding its row sum; row_bcast:15 / row_bcast:31 chain the four rows into
// lane 63, which is read into an SGPR (the result is wave-uniform).  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);  // row_mirror
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
