// swc_convnext64_mlp: the ConvNeXt block's MLP half (modules.py:1241-1247) on 64-frame tiles with TWO workgroups per CU.
//     x[M][C] (f32 residual stream)  +=  gamma * ( GELU( y W1^T + b1 ) W2^T + b2 ),      y = LayerNorm(dwconv7(x)) in bf16 (swc_dwconv7_ln)
// An experiment of round 4: swc_convnext_block runs one 128-frame workgroup per CU, one wave per SIMD with 512
// registers, and all 250 workgroups enter their HBM phases (front half, epilogue) together with the matrix pipe idle.  Here a
// workgroup owns 64 frames: 128 accumulator registers (AGPRs) + 128 VGPRs per wave, 80 KiB of LDS (y tile 64 KiB + H exchange
// 16 KiB), so two workgroups share a CU (2 waves per SIMD: one's stalls are the other's issue slots) and the second one can be
// started late (`stagger_cycles`) so that its HBM phases fall into the first one's slice loop.  The price is the weight stream:
// every 1 KiB fragment feeds two MFMAs instead of four.  Structure and operand maps as in swc_convnext.hip / swc_mlp.hip:
//   slice j of 128 hidden values: GEMM1 wave w: H^T[32 hidden][64 frames] = W1 . y^T (K = 512, 32 k-steps x 2 MFMAs, VGPR-form asm);
//   GELU on the accumulators, converted in place to bf16 B fragments, exchanged through LDS; GEMM2 wave w: out^T[128 columns]
//   [64 frames] += W2 . H^T (8 k-steps x 8 MFMAs, 128 AGPRs); weights global -> VGPR from a per-wave stream (8 fragments in flight).
#include <type_traits>
#include "swc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C6_C = 512;
constexpr int C6_BM = 64;
constexpr int C6_SL = 128;
constexpr int C6_PF = 8;
constexpr int C6_KS1 = C6_C / 16;   // 32
constexpr int C6_KS2 = C6_SL / 16;  // 8
constexpr int C6_NB = C6_C / 4 / 32;  // 4 column blocks per wave
constexpr int C6_FPP = 32;          // fragments per phase and wave: GEMM1 32 x 1, GEMM2 8 x 4
constexpr int C6_Y_BYTES = C6_BM * C6_C * 2;   // 64 KiB
constexpr int C6_H_BYTES = C6_SL * C6_BM * 2;  // 16 KiB
constexpr int C6_LDS = C6_Y_BYTES + C6_H_BYTES;  // 80 KiB: two workgroups per CU
constexpr int C6_TLD = C6_C + 4;
static_assert(32 * C6_TLD * 4 <= C6_LDS, "epilogue buffer");

__device__ __forceinline__ void c6_glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}

__device__ __forceinline__ unsigned c6_pack_bf16x2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ void c6_mfma1x2_vgpr(const u32x4& a, const u32x4& b0, const u32x4& b1, f32x16& c0, f32x16& c1) {
    asm("s_nop 1\n\t"
        "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
        "v_mfma_f32_32x32x16_bf16 %1, %2, %4, %1"
        : "+v"(c0), "+v"(c1)
        : "v"(a), "v"(b0), "v"(b1));
}

__device__ __forceinline__ f32x16 c6_mfma32(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b),
                                                   c, 0, 0, 0);
}

// wstream: per wave w (4 of them) NS * 64 + C6_PF fragments of 1 KiB in the order of consumption (convnext64_pack_kernel)
__global__ __launch_bounds__(256, 2) void convnext64_kernel(const bf16_t* __restrict__ y, const u32x4* __restrict__ wstream,
                                                           const float* __restrict__ b1, const float* __restrict__ b2,
                                                           const float* __restrict__ gamma, const float* x, float* xo, int M,
                                                           int NS, int stagger_cycles, int first_wave_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lf = lane & 31, lh = lane >> 5;
    const int row0 = blockIdx.x * C6_BM;
    // the second workgroup of a CU (the dispatcher hands out one workgroup per CU before it doubles up) starts late: its HBM
    // phases then fall into the first one's slice loop and vice versa
    if (stagger_cycles > 0 && (int)blockIdx.x >= first_wave_blocks) {
        const long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < stagger_cycles) __builtin_amdgcn_s_sleep(32);
    }

    // ---- y tile -> LDS as B fragments by DMA: fragment (s, fb) at [(2 s + fb)][lane][16 B]
    {
        const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int frag = w * 16 + i;
            const int s = frag >> 1, fb = frag & 1;
            int row = row0 + 32 * fb + lf;
            row = row < M ? row : M - 1;
            c6_glds16(y + (long)row * C6_C + 16 * s + 8 * lh, lds0 + frag * 1024);
        }
    }
    const u32x4* ylds = reinterpret_cast<const u32x4*>(smem) + lane;
    u32x4* hlds = reinterpret_cast<u32x4*>(smem + C6_Y_BYTES) + lane;

    const long per_wave = (long)NS * 64 + C6_PF;
    const char* wbase = reinterpret_cast<const char*>(wstream) + (long)w * per_wave * 1024;
    const unsigned lane_off = (unsigned)lane * 16u;
    auto wfrag = [&](int i) __attribute__((always_inline)) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(wbase + (long)i * 1024 + lane_off);
    };
    u32x4 ring[C6_PF];
#pragma unroll
    for (int i = 0; i < C6_PF; ++i) ring[i] = wfrag(i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc2[C6_NB][2];
#pragma unroll
    for (int a = 0; a < C6_NB; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
    f32x16 acc1[2];  // [frame block]: H^T tile of this wave's 32 hidden rows

    auto y_frags = [&](int s, u32x4 (&dst)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b) dst[b] = ylds[(s * 2 + b) * 64];
    };
    auto h_frags = [&](int q, u32x4 (&dst)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b) dst[b] = hlds[(q * 2 + b) * 64];
    };
    float4 bias_nx[4];
    auto load_bias = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_nx[g] = *reinterpret_cast<const float4*>(b1 + (long)j * C6_SL + 32 * w + 8 * g + 4 * lh);
    };
    load_bias(0);
    auto gemm1 = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[b][r] = reinterpret_cast<const float*>(&bias_nx[r >> 2])[r & 3];
        load_bias(j + 1 < NS ? j + 1 : 0);
        u32x4 yA[2], yB[2];
        y_frags(0, yA);
#pragma unroll
        for (int s = 0; s < C6_KS1; s += 2) {
            {
                y_frags(s + 1, yB);
                c6_mfma1x2_vgpr(ring[s % C6_PF], yA[0], yA[1], acc1[0], acc1[1]);
                ring[s % C6_PF] = wfrag(s + C6_PF);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                y_frags((s + 2) % C6_KS1, yA);
                c6_mfma1x2_vgpr(ring[(s + 1) % C6_PF], yB[0], yB[1], acc1[0], acc1[1]);
                ring[(s + 1) % C6_PF] = wfrag(s + 1 + C6_PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc1[0]), "+v"(acc1[1]));
        wbase += C6_FPP * 1024;
    };
    // GELU of accumulator registers 8 t + 4 h .. + 3 of tile b -> two packed dwords of B fragment t, in place
    auto gelu_half = [&](int b, int t, int h) __attribute__((always_inline)) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(acc1[b][8 * t + 4 * h + e]);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc1[b][4 * t + 2 * h + i] = __uint_as_float(c6_pack_bf16x2(v[2 * i], v[2 * i + 1]));
    };
    auto store_h = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                hlds[((2 * w + t) * 2 + b) * 64] = (u32x4){__float_as_uint(acc1[b][4 * t]), __float_as_uint(acc1[b][4 * t + 1]),
                                                           __float_as_uint(acc1[b][4 * t + 2]), __float_as_uint(acc1[b][4 * t + 3])};
    };
    auto gemm2 = [&](auto with_gelu) __attribute__((always_inline)) {
        u32x4 hA[2], hB[2];
        h_frags(0, hA);
        auto step = [&](int q, u32x4 (&cur)[2], u32x4 (&nxt)[2]) __attribute__((always_inline)) {
            if (q + 1 < C6_KS2) h_frags(q + 1, nxt);
#pragma unroll
            for (int n = 0; n < C6_NB; ++n) {
#pragma unroll
                for (int b = 0; b < 2; ++b) acc2[n][b] = c6_mfma32(ring[(q * C6_NB + n) % C6_PF], cur[b], acc2[n][b]);
                ring[(q * C6_NB + n) % C6_PF] = wfrag(q * C6_NB + n + C6_PF);
            }
            if constexpr (decltype(with_gelu)::value) gelu_half(q >> 2, (q >> 1) & 1, q & 1);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int q = 0; q < C6_KS2; q += 2) {
            step(q, hA, hB);
            step(q + 1, hB, hA);
        }
        wbase += C6_FPP * 1024;
    };

    gemm1(0);
#pragma unroll
    for (int q = 0; q < C6_KS2; ++q) gelu_half(q >> 2, (q >> 1) & 1, q & 1);
    store_h();
    __syncthreads();
    for (int j = 1; j < NS; ++j) {
        gemm1(j);
        gemm2(std::true_type{});
        __syncthreads();
        store_h();
        __syncthreads();
    }
    gemm2(std::false_type{});
    __syncthreads();

    // ---- epilogue: x[row][n] += gamma[n] * (out[row][n] + b2[n]) via a transposed f32 image [32 frames][516]
    float* tl = reinterpret_cast<float*>(smem);
    float4 g4[2], c4[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        g4[hf] = *reinterpret_cast<const float4*>(gamma + 256 * hf + 4 * lane);
        c4[hf] = *reinterpret_cast<const float4*>(b2 + 256 * hf + 4 * lane);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float4 rr[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long row = (long)row0 + 32 * p + 8 * w + i;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                rr[i][hf] = row < M ? *reinterpret_cast<const float4*>(x + row * C6_C + 256 * hf + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int n = 0; n < C6_NB; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& t = acc2[n][p];
                *reinterpret_cast<float4*>(tl + lf * C6_TLD + 32 * (C6_NB * w + n) + 8 * g + 4 * lh) =
                    make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int fl = 8 * w + i;
            const long row = (long)row0 + 32 * p + fl;
            if (row < M) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 v = *reinterpret_cast<const float4*>(tl + fl * C6_TLD + 256 * hf + 4 * lane);
                    float4 r = rr[i][hf];
                    r.x += g4[hf].x * (v.x + c4[hf].x); r.y += g4[hf].y * (v.y + c4[hf].y);
                    r.z += g4[hf].z * (v.z + c4[hf].z); r.w += g4[hf].w * (v.w + c4[hf].w);
                    *reinterpret_cast<float4*>(xo + row * C6_C + 256 * hf + 4 * lane) = r;
                }
            }
        }
        __syncthreads();
    }
}

// Stream of wave w: phases of 32 fragments G1(0), G1(1), G2(0), G1(2), G2(1), ..., G1(NS-1), G2(NS-2), G2(NS-1), then C6_PF zeros.
//   G1(j), fragment i: k-step i: W1 rows 128 j + 32 w + (lane & 31), columns 16 i + 8 (lane >> 5) .. + 7
//   G2(j), fragment i: k-step q = i / 4, column block n = i % 4: W2 rows 128 w + 32 n + (lane & 31), hidden values
//                      128 j + 16 q + 4 (lane >> 5) + {0..3, 8..11}
__global__ void convnext64_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, uint4* __restrict__ out, int NS) {
    const long per_wave = (long)NS * 64 + C6_PF;
    const long total = 4 * per_wave * 64;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int lane = (int)(id & 63);
    const long f_all = id >> 6;
    const int w = (int)(f_all / per_wave);
    const long f = f_all - (long)w * per_wave;
    const int lf = lane & 31, lh = lane >> 5;
    const long I = (long)NS * C6_SL;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (f < (long)NS * 64) {
        const int p = (int)(f >> 5), i = (int)(f & 31);
        const bool is_g1 = p == 0 || ((p & 1) && p < 2 * NS - 1);
        if (is_g1) {
            const int j = p == 0 ? 0 : (p + 1) >> 1;
            const long row = (long)j * C6_SL + 32 * w + lf;
            v = *reinterpret_cast<const uint4*>(w1 + row * C6_C + 16 * i + 8 * lh);
        } else {
            const int j = p == 2 * NS - 1 ? NS - 1 : (p >> 1) - 1;
            const int q = i >> 2, n = i & 3;
            const long nrow = 128 * w + 32 * n + lf;
            const long hid = (long)j * C6_SL + 16 * q + 4 * lh;
            const uint2 lo = *reinterpret_cast<const uint2*>(w2 + nrow * I + hid);
            const uint2 hi = *reinterpret_cast<const uint2*>(w2 + nrow * I + hid + 8);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    out[id] = v;
}

}  // namespace

extern "C" int64_t swc_convnext64_stream_bytes(int32_t C, int32_t I) {
    if (C != C6_C || I <= 0 || I % C6_SL != 0) return 0;
    return 4L * ((long)(I / C6_SL) * 64 + C6_PF) * 1024;
}

extern "C" int swc_convnext64_pack(const void* w1, const void* w2, void* stream_out, int32_t C, int32_t I, void* stream) {
    SWC_CHECK_ARG(w1 && w2 && stream_out, "swc_convnext64_pack: null pointer");
    SWC_CHECK_ARG(C == C6_C && I > 0 && I % C6_SL == 0, "swc_convnext64_pack: needs C = %d and I a multiple of %d (C=%d I=%d)", C6_C,
                  C6_SL, C, I);
    SWC_CHECK_ARG(aligned16(w1) && aligned16(w2) && aligned16(stream_out), "swc_convnext64_pack: unaligned");
    const int NS = I / C6_SL;
    const long total = 4L * ((long)NS * 64 + C6_PF) * 64;
    hipLaunchKernelGGL(convnext64_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w1, (const bf16_t*)w2, (uint4*)stream_out, NS);
    SWC_CHECK_LAUNCH("swc_convnext64_pack");
    return SWC_OK;
}

extern "C" int swc_convnext64_mlp(const void* y, const void* w_stream, const float* b1, const float* b2, const float* gamma,
                                  float* x, int32_t M, int32_t C, int32_t I, int32_t stagger_cycles, void* stream) {
    SWC_CHECK_ARG(y && w_stream && b1 && b2 && gamma && x, "swc_convnext64_mlp: null pointer");
    SWC_CHECK_ARG(C == C6_C && I > 0 && I % C6_SL == 0, "swc_convnext64_mlp: needs C = %d and I a multiple of %d (C=%d I=%d)", C6_C,
                  C6_SL, C, I);
    SWC_CHECK_ARG(M >= 0 && stagger_cycles >= 0, "swc_convnext64_mlp: bad M / stagger");
    SWC_CHECK_ARG(aligned16(y) && aligned16(w_stream) && aligned16(b1) && aligned16(b2) && aligned16(gamma) && aligned16(x),
                  "swc_convnext64_mlp: unaligned");
    if (M == 0) return SWC_OK;
    SWC_ENABLE_LDS(convnext64_kernel, C6_LDS, "swc_convnext64_mlp");
    const unsigned grid = (unsigned)((M + C6_BM - 1) / C6_BM);
    hipLaunchKernelGGL(convnext64_kernel, dim3(grid), dim3(256), C6_LDS, (hipStream_t)stream, (const bf16_t*)y,
                       (const u32x4*)w_stream, b1, b2, gamma, x, x, M, I / C6_SL, stagger_cycles, 256);
    SWC_CHECK_LAUNCH("swc_convnext64_mlp");
    return SWC_OK;
}
