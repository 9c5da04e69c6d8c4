// swc_convnext_mlp: the ConvNeXt block's MLP half in ONE kernel on gfx950 MFMA (modules.py:1241-1247):
//     x[M][C] (f32 residual stream)  +=  gamma * ( GELU( y W1^T + b1 ) W2^T + b2 ),      y = LayerNorm(dwconv7(x)) in bf16
// The [M][I] intermediate (I = 4096: 262 MB per block at M = 32000) never exists in memory.
//
// Decomposition (C = 512, 4 waves, one per SIMD, up to 512 registers each, one workgroup per CU):
//   * a workgroup owns 128 frames; its y tile sits in LDS for the whole kernel (128 KiB) as MFMA B-operand fragments;
//   * the hidden dimension is walked in slices of 128.  In slice j
//       GEMM1: wave w computes H^T[32 hidden of its own][128 frames] = W1[4j + w] . y^T        (K = 512)
//       GELU + bias on the accumulators, converted to bf16 IN REGISTERS: a 32x32 accumulator tile, rows pairwise
//              converted, IS the B operand of the next product (k order permuted; W2 is packed in that order),
//              then exchanged through a 32 KiB LDS buffer because every wave needs all 128 hidden values;
//       GEMM2: wave w accumulates out^T[128 n of its own][128 frames] += W2[n][slice j] . H^T  (256 accumulator registers);
//   * WEIGHTS NEVER TOUCH LDS: GEMM1 splits the hidden rows and GEMM2 the output columns over the waves, so every weight
//     fragment is needed by exactly one wave and goes global -> VGPR as one contiguous 1 KiB wave load.  The host packs
//     both matrices once (swc_convnext_pack) into the per-wave order of consumption: the kernel streams one pointer;
//   * software pipeline: GEMM1(j), then GEMM2(j-1) with the GELU of slice j spread over its 8 k-steps (VALU in the
//     MFMA shadow), barrier, H_j to LDS, barrier;
//   * epilogue: out^T through LDS (transposed, two passes of 64 frames) so that the residual stream is read and written
//     in whole 2 KiB rows.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  Operand maps (lane l): A[row l&31][k = 8(l>>5) + j], B[k = 8(l>>5) + j][col l&31],
// D[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31], r = 0..15.
#include <type_traits>
#include "swc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CX_C = 512;        // channels (K of GEMM1, N of GEMM2)
constexpr int CX_BM = 128;       // frames per workgroup
constexpr int CX_SL = 128;       // hidden values per slice
constexpr int CX_PF = 8;         // weight fragments in flight per wave (8 KiB)
constexpr int CX_Y_BYTES = CX_BM * CX_C * 2;  // 128 KiB
constexpr int CX_H_BYTES = CX_SL * CX_BM * 2; // 32 KiB
constexpr int CX_LDS = CX_Y_BYTES + CX_H_BYTES;
constexpr int CX_TLD = CX_C + 4;  // row pitch (floats) of the epilogue transpose buffer: conflict-free b128 writes

__device__ __forceinline__ void cx_glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b),
                                                   c, 0, 0, 0);
}

// wstream: per wave w (4 of them) NS * 64 + CX_PF fragments of 1 KiB in the order of consumption (swc_convnext_pack)
__global__ __launch_bounds__(256, 1) void convnext_mlp_kernel(const bf16_t* __restrict__ y, const uint4* __restrict__ wstream,
                                                             const float* __restrict__ b1, const float* __restrict__ b2,
                                                             const float* __restrict__ gamma, float* __restrict__ x, int M,
                                                             int NS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lf = lane & 31, lh = lane >> 5;
    const int row0 = blockIdx.x * CX_BM;

    // ---- y tile -> LDS as B fragments: fragment (s, fb) = k-step s (16 channels) x frame block fb (32 frames) at
    // [(4 s + fb)][lane][16 B]; lane l supplies frame 32 fb + (l & 31), channels 16 s + 8 (l >> 5) .. + 7.  LDS-DMA with a
    // per-lane source address writes exactly this lane-linear image.
    {
        const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int frag = w * 32 + i;
            const int s = frag >> 2, fb = frag & 3;
            int row = row0 + 32 * fb + lf;
            row = row < M ? row : M - 1;  // rows beyond M are computed on a copy of the last row and never stored
            cx_glds16(y + (long)row * CX_C + 16 * s + 8 * lh, lds0 + frag * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const uint4* ylds = reinterpret_cast<const uint4*>(smem) + lane;
    uint4* hlds = reinterpret_cast<uint4*>(smem + CX_Y_BYTES) + lane;

    // ---- weight stream of this wave
    const long per_wave = (long)NS * 64 + CX_PF;  // fragments
    const uint4* wp = wstream + ((long)w * per_wave) * 64 + lane;
    uint4 ring[CX_PF];
#pragma unroll
    for (int i = 0; i < CX_PF; ++i) ring[i] = wp[(long)i * 64];
    long fi = 0;  // index of the next fragment to consume; fragment fi + CX_PF is fetched when fi is consumed

    f32x16 acc2[4][4];  // [n block of this wave][frame block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
    f32x16 acc1[4];  // [frame block]: H^T tile of this wave's 32 hidden rows
    float4 bias[4];  // b1 of those rows: bias[g][e] belongs to accumulator register 4 g + e

    auto gemm1 = [&](int j) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[b][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias[g] = *reinterpret_cast<const float4*>(b1 + (long)j * CX_SL + 32 * w + 8 * g + 4 * lh);
#pragma nounroll
        for (int s0 = 0; s0 < 32; s0 += CX_PF) {  // the ring turns once per trip: static register indices inside
#pragma unroll
            for (int u = 0; u < CX_PF; ++u) {
                const uint4 a = ring[u];
                ring[u] = wp[(fi + s0 + u + CX_PF) * 64];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint4 yb = ylds[((s0 + u) * 4 + b) * 64];
                    acc1[b] = mfma32(a, yb, acc1[b]);
                }
            }
        }
        fi += 32;
    };
    // GELU(acc + bias) of accumulator registers 8 t .. 8 t + 7 of frame block b -> one packed B fragment (k-step t)
    auto gelu_frag = [&](int b, int t) -> uint4 {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * t + e;
            const float bb = reinterpret_cast<const float*>(&bias[r >> 2])[r & 3];
            v[e] = gelu_fast(acc1[b][r] + bb);
        }
        uint4 u;
        u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
        u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
        return u;
    };
    uint4 hp[4][2];  // packed GELU outputs of this wave: [frame block][k-step]
    auto store_h = [&]() {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) hlds[((2 * w + t) * 4 + b) * 64] = hp[b][t];
    };
    // GEMM2 over the slice whose H^T is in LDS; `with_gelu`: the GELU of the NEXT slice (acc1) rides along, one packed
    // fragment per k-step
    auto gemm2 = [&](auto with_gelu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint4 a[4], hb[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                a[n] = ring[(q * 4 + n) % CX_PF];
                ring[(q * 4 + n) % CX_PF] = wp[(fi + q * 4 + n + CX_PF) * 64];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) hb[b] = hlds[(q * 4 + b) * 64];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc2[n][b] = mfma32(a[n], hb[b], acc2[n][b]);
            if constexpr (decltype(with_gelu)::value) hp[q >> 1][q & 1] = gelu_frag(q >> 1, q & 1);
        }
        fi += 32;
    };

    // ---- pipeline over the hidden slices
    gemm1(0);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t) hp[b][t] = gelu_frag(b, t);
    store_h();
    __syncthreads();
    for (int j = 1; j < NS; ++j) {
        gemm1(j);
        gemm2(std::true_type{});
        __syncthreads();  // every wave has read H_{j-1}
        store_h();
        __syncthreads();  // H_j visible
    }
    gemm2(std::false_type{});
    __syncthreads();  // LDS is free: the epilogue re-uses all of it

    // ---- epilogue: x[row][n] += gamma[n] * (out[row][n] + b2[n]), via a transposed f32 image [64 frames][516]
    float* tl = reinterpret_cast<float*>(smem);
    float4 g4[2], c4[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        g4[hf] = *reinterpret_cast<const float4*>(gamma + 256 * hf + 4 * lane);
        c4[hf] = *reinterpret_cast<const float4*>(b2 + 256 * hf + 4 * lane);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int fbl = 0; fbl < 2; ++fbl)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& t = acc2[n][2 * p + fbl];
                    *reinterpret_cast<float4*>(tl + (32 * fbl + lf) * CX_TLD + 32 * (4 * w + n) + 8 * g + 4 * lh) =
                        make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
                }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int fl = 16 * w + i;
            const long row = (long)row0 + 64 * p + fl;
            if (row < M) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 v = *reinterpret_cast<const float4*>(tl + fl * CX_TLD + 256 * hf + 4 * lane);
                    float4* xp = reinterpret_cast<float4*>(x + row * CX_C + 256 * hf + 4 * lane);
                    float4 r = *xp;
                    r.x += g4[hf].x * (v.x + c4[hf].x); r.y += g4[hf].y * (v.y + c4[hf].y);
                    r.z += g4[hf].z * (v.z + c4[hf].z); r.w += g4[hf].w * (v.w + c4[hf].w);
                    *xp = r;
                }
            }
        }
        __syncthreads();
    }
}

// One thread per 16-byte chunk of the packed stream.  Stream of wave w: for slice j: [GEMM1(j) fragments: W1 rows
// 128 j + 32 w .. + 31, k-steps s = 0..31], and after GEMM1(j) for j >= 1 (and once more at the end) the GEMM2 fragments
// of slice j - 1: for k-step q = 0..7, n block 4 w + n, n = 0..3.  Consumption order: G1(0), G1(1), G2(0), G1(2), G2(1), ...
__global__ void convnext_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, uint4* __restrict__ out,
                                     int NS) {
    const long per_wave = (long)NS * 64 + CX_PF;
    const long total = 4 * per_wave * 64;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int lane = (int)(id & 63);
    const long f_all = id >> 6;
    const int w = (int)(f_all / per_wave);
    const long f = f_all - (long)w * per_wave;
    const int lf = lane & 31, lh = lane >> 5;
    const long I = (long)NS * CX_SL;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (f < (long)NS * 64) {
        // phase p: 0 -> G1(0); 2k-1 -> G1(k), 2k -> G2(k-1) for k = 1..NS-1; 2NS-1 -> G2(NS-1)
        const int p = (int)(f >> 5), i = (int)(f & 31);
        const bool is_g1 = p == 0 || ((p & 1) && p < 2 * NS - 1);
        if (is_g1) {
            const int j = p == 0 ? 0 : (p + 1) >> 1;
            const long row = (long)j * CX_SL + 32 * w + lf;  // hidden row
            v = *reinterpret_cast<const uint4*>(w1 + row * CX_C + 16 * i + 8 * lh);
        } else {
            const int j = p == 2 * NS - 1 ? NS - 1 : (p >> 1) - 1;
            const int q = i >> 2, n = i & 3;
            const long nrow = 32 * (4 * w + n) + lf;  // output column n of the block = row of W2
            const long hid = (long)j * CX_SL + 32 * (q >> 1) + 16 * (q & 1) + 4 * lh;  // elements jj: hid + 8 (jj >> 2) + (jj & 3)
            const uint2 lo = *reinterpret_cast<const uint2*>(w2 + nrow * I + hid);
            const uint2 hi = *reinterpret_cast<const uint2*>(w2 + nrow * I + hid + 8);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    out[id] = v;
}

}  // namespace

extern "C" int64_t swc_convnext_stream_bytes(int32_t C, int32_t I) {
    if (C != CX_C || I <= 0 || I % CX_SL != 0) return 0;
    return 4L * ((long)(I / CX_SL) * 64 + CX_PF) * 1024;
}

extern "C" int swc_convnext_pack(const void* w1, const void* w2, void* stream_out, int32_t C, int32_t I, void* stream) {
    SWC_CHECK_ARG(w1 && w2 && stream_out, "swc_convnext_pack: null pointer");
    SWC_CHECK_ARG(C == CX_C && I > 0 && I % CX_SL == 0, "swc_convnext_pack: needs C = %d and I a multiple of %d (C=%d I=%d)",
                  CX_C, CX_SL, C, I);
    SWC_CHECK_ARG(aligned16(w1) && aligned16(w2) && aligned16(stream_out), "swc_convnext_pack: unaligned");
    const int NS = I / CX_SL;
    const long total = 4L * ((long)NS * 64 + CX_PF) * 64;
    hipLaunchKernelGGL(convnext_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w1, (const bf16_t*)w2, (uint4*)stream_out, NS);
    SWC_CHECK_LAUNCH("swc_convnext_pack");
    return SWC_OK;
}

extern "C" int swc_convnext_mlp(const void* y, const void* w_stream, const float* b1, const float* b2, const float* gamma,
                                float* x, int32_t M, int32_t C, int32_t I, void* stream) {
    SWC_CHECK_ARG(y && w_stream && b1 && b2 && gamma && x, "swc_convnext_mlp: null pointer");
    SWC_CHECK_ARG(C == CX_C && I > 0 && I % CX_SL == 0, "swc_convnext_mlp: needs C = %d and I a multiple of %d (C=%d I=%d)",
                  CX_C, CX_SL, C, I);
    SWC_CHECK_ARG(M >= 0, "swc_convnext_mlp: bad M");
    SWC_CHECK_ARG(aligned16(y) && aligned16(w_stream) && aligned16(b1) && aligned16(b2) && aligned16(gamma) && aligned16(x),
                  "swc_convnext_mlp: unaligned");
    if (M == 0) return SWC_OK;
    auto kern = convnext_mlp_kernel;
    SWC_ENABLE_LDS(kern, CX_LDS, "swc_convnext_mlp");
    const unsigned grid = (unsigned)((M + CX_BM - 1) / CX_BM);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), CX_LDS, (hipStream_t)stream, (const bf16_t*)y, (const uint4*)w_stream,
                       b1, b2, gamma, x, M, I / CX_SL);
    SWC_CHECK_LAUNCH("swc_convnext_mlp");
    return SWC_OK;
}
