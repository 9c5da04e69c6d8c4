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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // one MFMA operand fragment (8 bf16) as a native vector: asm "v" operand

constexpr int CX_C = 512;        // channels (K of GEMM1, N of GEMM2)
constexpr int CX_BM = 128;       // frames per workgroup
constexpr int CX_SL = 128;       // hidden values per slice
constexpr int CX_PF = 8;         // weight fragments in flight per wave (8 KiB)
constexpr int CX_Y_BYTES = CX_BM * CX_C * 2;  // 128 KiB
constexpr int CX_H_BYTES = CX_SL * CX_BM * 2; // 32 KiB
constexpr int CX_LDS = CX_Y_BYTES + CX_H_BYTES;
constexpr int CX_TLD = CX_C + 4;  // row pitch (floats) of the epilogue transpose buffer: conflict-free b128 writes
// Timing ablations for tuning builds only (-DCX_ABL=mask, wrong results): 1 = no barriers in the slice loop, 2 = no GELU
// arithmetic (accumulators are packed as they are), 4 = no weight loads in the loop (the ring keeps its first fragments),
// 8 = LDS fragment reads only in the first k-step of a phase, 16 = no epilogue
#ifndef CX_ABL
#define CX_ABL 0
#endif
#ifndef CX_MFMA16
#define CX_MFMA16 0   // build option: 1 = the v_mfma_f32_16x16x32_bf16 kernel (convnext16_kernel) instead of the 32 x 32 x 16 one
#endif
#ifndef CX_EPI_PREFETCH
#define CX_EPI_PREFETCH 1
#endif
// CX_RES_ACC (round 4, build option for the 32 x 32 x 16 kernel; measured, parity-green, NOT the default): the residual lives in
// the accumulators.  gamma is folded into W2 at pack time (swc_convnext_pack) and GEMM2's accumulators START at x + gamma * b2
// (loaded in the accumulator layout right after the front half), so the block's output IS the accumulator and the epilogue
// neither reads nor multiplies.  profiles/r04_convnext_residual_in_accumulators.txt: fabric traffic 278.5 -> 267.4 MB per launch
// only (the epilogue's second read of x was served by L2 already; the rest of the "2.0 x" is the weight stream fetched once per
// XCD and the halo rows), block 261.9 -> 263.2 us, MLP-only form 243.2 -> 251.2 us (48 strided loads per lane exposed behind the
// front half), whole step 20.42 -> 20.44 ms.
#ifndef CX_RES_ACC
#define CX_RES_ACC 0
#endif

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
    // one v_cvt_pk_bf16_f32 for the pair (RNE).  Only for operands that a plain VALU instruction produced (here: the fma
    // that ends gelu_fast): hipcc does not insert, for an asm statement, the wait state that a transcendental result needs
    // before its first use (swc_attention16 uses the vector-conversion form for that reason).
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// F16 (operand_dtype SWC_F16): the block's internal operands — LayerNorm output, GELU output, both weight matrices — are IEEE
// half precision instead of bf16: 11 significand bits instead of 8 at the same MFMA rate.  Their ranges are bounded by the block
// itself (|LayerNorm output| <= sqrt(C) |ln_w| + |ln_b|; hidden activations of O(10); weights < 1), three to four orders of
// magnitude inside +-65504; the conversion is RNE like the bf16 one.
template <bool F16>
__device__ __forceinline__ unsigned pack_x2(float lo, float hi) {
    if constexpr (F16) {
        unsigned r;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    } else {
        return pack_bf16x2(lo, hi);
    }
}

// GEMM1's MFMAs in VGPR form.  The 256 accumulators of GEMM2 fill the AGPR half of the register file; left to hipcc, the
// 64 accumulators of GEMM1 are also given AGPR-form MFMAs and the two sets are shuffled between the halves with ~1500
// v_accvgpr_read/write/mov per slice (6 k issue cycles beside 8 k MFMA cycles).  Written as asm with "v" operands these
// four MFMAs keep their accumulators in VGPRs, where the GELU reads them directly.  One statement per k-step: the
// leading s_nop 1 covers a VALU copy of an operand hipcc may have placed right in front (it pads nothing inside asm).
template <bool F16 = false>
__device__ __forceinline__ void mfma32x4_vgpr(const u32x4& a, const u32x4& b0, const u32x4& b1, const u32x4& b2, const u32x4& b3,
                                              f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3) {
    if constexpr (F16)
        asm("s_nop 1\n\t"
            "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\t"
            "v_mfma_f32_32x32x16_f16 %1, %4, %6, %1\n\t"
            "v_mfma_f32_32x32x16_f16 %2, %4, %7, %2\n\t"
            "v_mfma_f32_32x32x16_f16 %3, %4, %8, %3"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
            : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    else
        asm("s_nop 1\n\t"
            "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %4, %6, %1\n\t"
            "v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n\t"
            "v_mfma_f32_32x32x16_bf16 %3, %4, %8, %3"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
            : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

template <bool F16 = false>
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b),
                                                       c, 0, 0, 0);
}

// front half of the block for the fused form: depthwise Conv1d(k7, pad 3, per utterance) + LayerNorm (modules.py:1233-1239)
struct CxFront {
    const float* w7;    // [7][C] depthwise taps
    const float* bias;  // [C]
    const float* ln_w;  // [C]
    const float* ln_b;  // [C]
    int T;              // frames per utterance: rows b * T + t; taps do not cross utterances (zero padding)
    float eps;
    const int* t_limit;  // optional [B]: frames t >= t_limit[b] of utterance b need not be computed (tiles wholly beyond are skipped)
};

// wstream: per wave w (4 of them) NS * 64 + CX_PF fragments of 1 KiB in the order of consumption (swc_convnext_pack)
// FUSED_DW: y is not read; the workgroup computes LayerNorm(dwconv7(x)) of its 128 frames itself (front half above)
template <bool FUSED_DW, bool F16 = false>
__global__ __launch_bounds__(256, 1) void convnext_mlp_kernel(const bf16_t* __restrict__ y, const u32x4* __restrict__ wstream,
                                                             const float* __restrict__ b1, const float* __restrict__ b2,
                                                             const float* __restrict__ gamma, const float* x, float* xo,
                                                             int M, int NS, CxFront fr) {
    // x: residual stream in (front half incl. halo rows of the neighbouring tiles, and the residual add); xo: residual
    // stream out.  The fused form must not run in place: a later tile would read halo rows an earlier tile has updated.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lf = lane & 31, lh = lane >> 5;
    const int row0 = blockIdx.x * CX_BM;
    if constexpr (FUSED_DW) {
        // Ragged batches: a tile whose frames all lie at or beyond their utterances' limits does nothing (its rows of x_out
        // stay undefined: the caller's limits include the receptive field of everything it keeps).  Block-uniform.
        if (fr.t_limit) {
            const int last = (row0 + CX_BM - 1 < M ? row0 + CX_BM - 1 : M - 1);
            bool need = false;
            for (int b = row0 / fr.T; b <= last / fr.T; ++b) {
                const int t_lo = (row0 > b * fr.T ? row0 : b * fr.T) - b * fr.T;
                need = need || t_lo < fr.t_limit[b];
            }
            if (!need) return;
        }
    }

    // ---- y tile -> LDS as B fragments: fragment (s, fb) = k-step s (16 channels) x frame block fb (32 frames) at
    // [(4 s + fb)][lane][16 B]; lane l supplies frame 32 fb + (l & 31), channels 16 s + 8 (l >> 5) .. + 7.  LDS-DMA with a
    // per-lane source address writes exactly this lane-linear image.
    if constexpr (!FUSED_DW) {
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
    } else {
        // Wave w turns out the 32 frames of frame block w: a lane owns channels 4 l .. 4 l + 3 and 256 + 4 l .. + 3 of
        // every row (whole 2 KiB rows per wave load, LayerNorm sums on the DPP path, nothing crosses waves).  Groups of 4
        // frames: their 10 input rows sit in registers, the 4 new rows of the next group are in flight meanwhile.
        // Same arithmetic, in the same order, as swc_dwconv7_ln (bias, taps 0..6 as fma, two-pass LayerNorm).
        const float4* x4 = reinterpret_cast<const float4*>(x);
        // the 7 x 8 taps stay in registers; bias and the LayerNorm affine (used once per group) are re-read from the cache
        // per group so that TWO groups of new rows can be in flight (the rows come from beyond L2: one group ahead left
        // most of their latency exposed, 26 us per block against 21 us for the stand-alone kernel)
        float4 wr[7][2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 7; ++j) wr[j][k] = reinterpret_cast<const float4*>(fr.w7 + (long)j * CX_C)[lane + 64 * k];
        const int f0w = row0 + 32 * w;  // first row of this wave
        // Waves whose 38-row window lies inside one utterance and inside the tensor (all but one in ~8 at T = 1000) skip
        // every boundary test: the tests are integer divisions on the scalar unit, 54 per group
        const bool interior = f0w - 3 >= 0 && f0w + 35 < M && (f0w - 3) / fr.T == (f0w + 35) / fr.T;
        auto load_row = [&](int r, float4 (&dst)[2]) {  // rows outside the tensor are zeros (wave-uniform test)
            const bool ok = interior || (r >= 0 && r < M);
#pragma unroll
            for (int k = 0; k < 2; ++k) dst[k] = ok ? x4[(long)r * (CX_C / 4) + lane + 64 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // A row stays in the window for up to three groups and can be a valid tap for one frame and lie beyond an utterance
        // boundary for another: rows are loaded as they are, the utterance test is made per (frame, row) pair when used.
        float4 win[10][2], nxA[4][2], nxB[4][2];
#pragma unroll
        for (int p = 0; p < 10; ++p) load_row(f0w - 3 + p, win[p]);
#pragma unroll
        for (int p = 0; p < 4; ++p) load_row(f0w + 4 + 3 + p, nxA[p]);  // the new rows of group 1
        const float inv_c = 1.0f / (float)CX_C;
        char* ybase = smem;
        auto group = [&](int g, float4 (&cur)[4][2], float4 (&pre)[4][2]) {
            // `cur`: the 4 new rows of group g + 1 (already in flight); `pre`: where those of group g + 2 are loaded now
            if (g + 2 < 8) {
#pragma unroll
                for (int p = 0; p < 4; ++p) load_row(f0w + 4 * (g + 2) + 3 + p, pre[p]);
            }
            float4 br[2], gw[2], gb[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                br[k] = reinterpret_cast<const float4*>(fr.bias)[lane + 64 * k];
                gw[k] = reinterpret_cast<const float4*>(fr.ln_w)[lane + 64 * k];
                gb[k] = reinterpret_cast<const float4*>(fr.ln_b)[lane + 64 * k];
            }
            int ur[10], uf[4];  // utterance of every window row / frame of the group (boundary waves only)
            if (!interior) {
#pragma unroll
                for (int p = 0; p < 10; ++p) {
                    const int r = f0w + 4 * g - 3 + p;
                    ur[p] = r < 0 ? -1 : r / fr.T;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) uf[u] = ur[u + 3];
            }
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 2; ++k) v[u][k] = br[k];
#pragma unroll
            for (int p = 0; p < 10; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = p - u;  // frame u takes window row p as tap j
                    if (j >= 0 && j < 7) {
                        const bool ok = interior || ur[p] == uf[u];  // same utterance (wave-uniform)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float4 xv = ok ? win[p][k] : make_float4(0.f, 0.f, 0.f, 0.f);
                            v[u][k].x += xv.x * wr[j][k].x; v[u][k].y += xv.y * wr[j][k].y;
                            v[u][k].z += xv.z * wr[j][k].z; v[u][k].w += xv.w * wr[j][k].w;
                        }
                    }
                }
            float sum[4], sq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sum[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) sum[u] += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum[u] = wave_sum_dpp(sum[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float mean = sum[u] * inv_c;
                sq[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    v[u][k].x -= mean; v[u][k].y -= mean; v[u][k].z -= mean; v[u][k].w -= mean;
                    sq[u] += (v[u][k].x * v[u][k].x + v[u][k].y * v[u][k].y) + (v[u][k].z * v[u][k].z + v[u][k].w * v[u][k].w);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sq[u] = wave_sum_dpp(sq[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float rstd = rsqrtf(sq[u] * inv_c + fr.eps);
                const int fl = 4 * g + u;  // frame inside this wave's block
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float o0 = v[u][k].x * rstd * gw[k].x + gb[k].x, o1 = v[u][k].y * rstd * gw[k].y + gb[k].y;
                    const float o2 = v[u][k].z * rstd * gw[k].z + gb[k].z, o3 = v[u][k].w * rstd * gw[k].w + gb[k].w;
                    // channels c = 256 k + 4 l .. + 3 -> fragment (s = c / 16, fb = w), lane' = 32 ((c % 16) / 8) + frame,
                    // byte (c % 8) * 2
                    const int s_ = 16 * k + (lane >> 2);
                    const int off = ((s_ * 4 + w) * 64 + 32 * ((lane >> 1) & 1) + fl) * 16 + (lane & 1) * 8;
                    *reinterpret_cast<uint2*>(ybase + off) = make_uint2(pack_x2<F16>(o0, o1), pack_x2<F16>(o2, o3));
                }
            }
            // slide the window by 4 rows
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) win[p][k] = win[p + 4][k];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) win[6 + p][k] = cur[p][k];
        };
        for (int g = 0; g < 8; g += 2) {
            group(g, nxA, nxB);
            group(g + 1, nxB, nxA);
        }
        __syncthreads();
    }
    const u32x4* ylds = reinterpret_cast<const u32x4*>(smem) + lane;
    u32x4* hlds = reinterpret_cast<u32x4*>(smem + CX_Y_BYTES) + lane;

    // ---- weight stream of this wave
    // address = wave-uniform byte pointer (SGPR pair, advanced once per phase) + one 32-bit lane offset + immediate:
    // per-fragment 64-bit VGPR addresses cost 28 registers and a spill in the slice loop
    const long per_wave = (long)NS * 64 + CX_PF;  // fragments
    const char* wbase = reinterpret_cast<const char*>(wstream) + (long)w * per_wave * 1024;
    const unsigned lane_off = (unsigned)lane * 16u;
    auto wfrag = [&](int i) -> u32x4 {  // fragment i of the current phase (i may run CX_PF past its end)
        if (CX_ABL & 4) i &= CX_PF - 1;
        return *reinterpret_cast<const u32x4*>(wbase + (long)i * 1024 + lane_off);
    };
    u32x4 ring[CX_PF];
#pragma unroll
    for (int i = 0; i < CX_PF; ++i) ring[i] = wfrag(i);

    f32x16 acc2[4][4];  // [n block of this wave][frame block]
#if CX_RES_ACC
    // the accumulators start at the residual stream + gamma * b2: register r = 4 g + e of tile (n, b) in lane (lf, lh) is frame
    // 32 b + lf, column 128 w + 32 n + 8 g + 4 lh + e.  One scheduling region per column block: 16 strided 16-byte loads each
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 128 * w + 32 * a + 8 * g + 4 * lh;
            const float4 gv = *reinterpret_cast<const float4*>(gamma + col);
            const float4 bv = *reinterpret_cast<const float4*>(b2 + col);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long row = (long)row0 + 32 * b + lf;
                const float4 xv = row < M ? *reinterpret_cast<const float4*>(x + row * CX_C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc2[a][b][4 * g] = fmaf(gv.x, bv.x, xv.x); acc2[a][b][4 * g + 1] = fmaf(gv.y, bv.y, xv.y);
                acc2[a][b][4 * g + 2] = fmaf(gv.z, bv.z, xv.z); acc2[a][b][4 * g + 3] = fmaf(gv.w, bv.w, xv.w);
            }
        }
        // the tiles are materialised in AGPRs HERE: left alone, hipcc sinks the adds and the v_accvgpr_writes to the first use
        // of the accumulators (GEMM2 of the slice loop) and carries the 64 loaded float4 through GEMM1(0) in VGPRs (66 spills)
        asm volatile("" : "+a"(acc2[a][0]), "+a"(acc2[a][1]), "+a"(acc2[a][2]), "+a"(acc2[a][3]));
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
#endif
    f32x16 acc1[4];  // [frame block]: H^T tile of this wave's 32 hidden rows

    // Every k-step is its own scheduling region (sched_barrier at its end): the step issues the refill of the ring slot it
    // consumes and the LDS reads of the NEXT step's B fragments, then its MFMAs.  Left to itself hipcc sinks the ring
    // loads to the end of the unrolled trip (issue -> use distance of two MFMAs instead of eight steps) and reads each B
    // fragment right in front of its MFMA (LDS latency exposed on every pair, one wave per SIMD has nobody to hide it).
    auto y_frags = [&](int s, u32x4 (&dst)[4]) {
        if ((CX_ABL & 8) && s > 1) return;
#pragma unroll
        for (int b = 0; b < 4; ++b) dst[b] = ylds[(s * 4 + b) * 64];
    };
    auto h_frags = [&](int q, u32x4 (&dst)[4]) {
        if ((CX_ABL & 8) && q > 1) return;
#pragma unroll
        for (int b = 0; b < 4; ++b) dst[b] = hlds[(q * 4 + b) * 64];
    };
    // b1 of this wave's 32 hidden rows, one slice ahead: register r = 4 g + e of a lane in half lh belongs to row
    // 8 g + 4 lh + e.  Loaded a whole slice before its use, so waiting for it never drains the younger ring loads
    float4 bias_nx[4];
    auto load_bias = [&](int j) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias_nx[g] = *reinterpret_cast<const float4*>(b1 + (long)j * CX_SL + 32 * w + 8 * g + 4 * lh);
    };
    load_bias(0);
    auto gemm1 = [&](int j) {
        // accumulators start at the bias
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[b][r] = reinterpret_cast<const float*>(&bias_nx[r >> 2])[r & 3];
        load_bias(j + 1 < NS ? j + 1 : 0);
        u32x4 yA[4], yB[4];
        y_frags(0, yA);
        // fully unrolled: in a rolled loop the ring is loop-carried and hipcc copies all eight slots at the loop head,
        // which waits for (nearly) every load in flight and cuts the prefetch distance to one or two steps
#pragma unroll
        for (int s0 = 0; s0 < 32; s0 += CX_PF) {
#pragma unroll
            for (int u = 0; u < CX_PF; u += 2) {
                {
                    y_frags(s0 + u + 1, yB);
                    mfma32x4_vgpr<F16>(ring[u], yA[0], yA[1], yA[2], yA[3], acc1[0], acc1[1], acc1[2], acc1[3]);
                    ring[u] = wfrag(s0 + u + CX_PF);  // refill the slot just consumed (no copy of the operand)
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    y_frags((s0 + u + 2) & 31, yA);  // the last step re-reads step 0 (harmless)
                    mfma32x4_vgpr<F16>(ring[u + 1], yB[0], yB[1], yB[2], yB[3], acc1[0], acc1[1], acc1[2], acc1[3]);
                    ring[u + 1] = wfrag(s0 + u + 1 + CX_PF);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // MFMA results in VGPRs -> VALU readers: the wait states hipcc would insert for its own MFMAs (asm is opaque to it)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]));
        wbase += 32 * 1024;
    };
    // GELU of accumulator registers 8 t .. 8 t + 7 of frame block b -> one packed B fragment (k-step t), kept IN PLACE:
    // its four dwords replace registers 4 t .. 4 t + 3 of the same tile (already consumed), so the packed copy of the
    // slice costs no registers of its own
    auto gelu_frag = [&](int b, int t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (CX_ABL & 2) ? acc1[b][8 * t + e] : gelu_fast(acc1[b][8 * t + e]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[b][4 * t + i] = __uint_as_float(pack_x2<F16>(v[2 * i], v[2 * i + 1]));
    };
    auto store_h = [&]() {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                hlds[((2 * w + t) * 4 + b) * 64] =
                    (u32x4){__float_as_uint(acc1[b][4 * t]), __float_as_uint(acc1[b][4 * t + 1]),
                            __float_as_uint(acc1[b][4 * t + 2]), __float_as_uint(acc1[b][4 * t + 3])};
    };
    // GEMM2 over the slice whose H^T is in LDS; `with_gelu`: the GELU of the NEXT slice (acc1) rides along, one packed
    // fragment per k-step, in the same scheduling region as that step's 16 MFMAs
    auto gemm2 = [&](auto with_gelu) {
        u32x4 hA[4], hB[4];
        h_frags(0, hA);
        auto step = [&](int q, u32x4 (&cur)[4], u32x4 (&nxt)[4]) {
            if (q + 1 < 8) h_frags(q + 1, nxt);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int b = 0; b < 4; ++b) acc2[n][b] = mfma32<F16>(ring[(q * 4 + n) % CX_PF], cur[b], acc2[n][b]);
                ring[(q * 4 + n) % CX_PF] = wfrag(q * 4 + n + CX_PF);
            }
            if constexpr (decltype(with_gelu)::value) gelu_frag(q >> 1, q & 1);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            step(q, hA, hB);
            step(q + 1, hB, hA);
        }
        wbase += 32 * 1024;
    };

    // ---- pipeline over the hidden slices
    gemm1(0);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t) gelu_frag(b, t);
    store_h();
    __syncthreads();
    for (int j = 1; j < NS; ++j) {
        gemm1(j);
        gemm2(std::true_type{});
        if (!(CX_ABL & 1)) __syncthreads();  // every wave has read H_{j-1}
        store_h();
        if (!(CX_ABL & 1)) __syncthreads();  // H_j visible
    }
    gemm2(std::false_type{});
    __syncthreads();  // LDS is free: the epilogue re-uses all of it

    if (CX_ABL & 16) {  // keep the accumulators alive, store nothing
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) keep += acc2[a][b][0] + acc2[a][b][7];
        if (keep == 12345.678f) xo[0] = keep;
        return;
    }
    // ---- epilogue: x[row][n] += gamma[n] * (out[row][n] + b2[n]), via a transposed f32 image [64 frames][516]
    // (CX_RES_ACC: the accumulators hold exactly that already; the rows are only transposed and stored)
    float* tl = reinterpret_cast<float*>(smem);
    float4 g4[2], c4[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        g4[hf] = *reinterpret_cast<const float4*>(gamma + 256 * hf + 4 * lane);
        c4[hf] = *reinterpret_cast<const float4*>(b2 + 256 * hf + 4 * lane);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#if CX_EPI_PREFETCH && !CX_RES_ACC
        // the residual rows of this pass first: 32 independent 16-byte loads per lane, in flight across the LDS round trip
        // (the arithmetic registers of the slice loop are free here)
        float4 rr[16][2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long row = (long)row0 + 64 * p + 16 * w + i;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                rr[i][hf] = row < M ? *reinterpret_cast<const float4*>(x + row * CX_C + 256 * hf + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#endif
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
#if CX_EPI_PREFETCH
#pragma unroll
#else
#pragma unroll 4
#endif
        for (int i = 0; i < 16; ++i) {
            const int fl = 16 * w + i;
            const long row = (long)row0 + 64 * p + fl;
            if (row < M) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 v = *reinterpret_cast<const float4*>(tl + fl * CX_TLD + 256 * hf + 4 * lane);
#if CX_RES_ACC
                    const float4 r = v;
#else
#if CX_EPI_PREFETCH
                    float4 r = rr[i][hf];
#else
                    float4 r = *reinterpret_cast<const float4*>(x + row * CX_C + 256 * hf + 4 * lane);
#endif
                    r.x += g4[hf].x * (v.x + c4[hf].x); r.y += g4[hf].y * (v.y + c4[hf].y);
                    r.z += g4[hf].z * (v.z + c4[hf].z); r.w += g4[hf].w * (v.w + c4[hf].w);
#endif
                    *reinterpret_cast<float4*>(xo + row * CX_C + 256 * hf + 4 * lane) = r;
                }
            }
        }
        __syncthreads();
    }
}

#if CX_MFMA16
// GEMM1's MFMAs of half a 32-deep k-step in VGPR form (see mfma32x4_vgpr): 2 hidden tiles x 4 frame blocks
__device__ __forceinline__ void mfma16x8_vgpr(const u32x4& a0, const u32x4& a1, const u32x4 (&b)[4], f32x4& c00, f32x4& c01,
                                              f32x4& c02, f32x4& c03, f32x4& c10, f32x4& c11, f32x4& c12, f32x4& c13) {
    asm("s_nop 1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %8, %10, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %4, %9, %10, %4\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %8, %11, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %5, %9, %11, %5\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %8, %12, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %6, %9, %12, %6\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %8, %13, %3\n\t"
        "v_mfma_f32_16x16x32_bf16 %7, %9, %13, %7"
        : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13)
        : "v"(a0), "v"(a1), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// GEMM2's MFMAs of one n tile x 4 frame blocks with the accumulators pinned in AGPRs, in place.  The 64 accumulator tiles fill
// the AGPR file exactly; given the builtin, hipcc allocates destination != source for most of them and moves ~620 registers
// per slice through v_accvgpr_read / write / mov (measured: 279 us per block against 262 us for the 32 x 32 x 16 kernel).
__device__ __forceinline__ void mfma16x4_agpr(const u32x4& a, const u32x4 (&b)[4], f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    asm("s_nop 1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\t"
        "v_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"
        : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
        : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_bf16 (build option -DCX_MFMA16=1; parity-tested, not the default): the chip holds a
// higher clock on the 16 x 16 x 32 shape than on 32 x 32 x 16 at equal flops (tools/probes/mfma_valu_overlap.hip, random bf16
// operands: 128.6 against 151.7 ns per 262 kflop and SIMD, 2.05 against 1.69 GHz).  Measured (profiles/r03_convnext_mfma16.txt):
// on random operands from cold buffers 256.9 us per block against 266.4 us; inside the bench step, on the synthetic
// checkpoint's activations, 261 us against 249 us — there the 32 x 32 x 16 kernel is not held back by the clock and the halved
// vector-issue room behind a 16-cycle MFMA costs more than the shape gains.  Operand maps (lane l): A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][col l & 15],
// D[row 4 (l >> 4) + r][col l & 15], r = 0..3.  Per wave and slice: GEMM1 = 2 hidden tiles x 8 frame blocks (64 accumulator
// registers), 16 k-steps of 32 channels; the two hidden tiles of a frame block, converted pairwise, are ONE B fragment of
// GEMM2 (operand k index 8 h + j <-> hidden row 4 h + j for j < 4, 16 + 4 h + j - 4 otherwise; W2 is packed in that order);
// GEMM2 = 8 n tiles x 8 frame blocks (256 accumulator registers), 4 k-steps of 32 hidden values (one per producing wave).
// wstream: per wave w (4 of them) NS * 64 + CX_PF fragments of 1 KiB in the order of consumption (convnext16_pack_kernel)
// FUSED_DW: y is not read; the workgroup computes LayerNorm(dwconv7(x)) of its 128 frames itself (front half above)
template <bool FUSED_DW>
__global__ __launch_bounds__(256, 1) void convnext16_kernel(const bf16_t* __restrict__ y, const u32x4* __restrict__ wstream,
                                                             const float* __restrict__ b1, const float* __restrict__ b2,
                                                             const float* __restrict__ gamma, const float* x, float* xo,
                                                             int M, int NS, CxFront fr) {
    // x: residual stream in (front half incl. halo rows of the neighbouring tiles, and the residual add); xo: residual
    // stream out.  The fused form must not run in place: a later tile would read halo rows an earlier tile has updated.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lf = lane & 15, lh = lane >> 4;
    const int row0 = blockIdx.x * CX_BM;
    if constexpr (FUSED_DW) {
        // Ragged batches: a tile whose frames all lie at or beyond their utterances' limits does nothing (its rows of x_out
        // stay undefined: the caller's limits include the receptive field of everything it keeps).  Block-uniform.
        if (fr.t_limit) {
            const int last = (row0 + CX_BM - 1 < M ? row0 + CX_BM - 1 : M - 1);
            bool need = false;
            for (int b = row0 / fr.T; b <= last / fr.T; ++b) {
                const int t_lo = (row0 > b * fr.T ? row0 : b * fr.T) - b * fr.T;
                need = need || t_lo < fr.t_limit[b];
            }
            if (!need) return;
        }
    }

    // ---- y tile -> LDS as B fragments: fragment (s, fb) = k-step s (32 channels) x frame block fb (16 frames) at
    // [(8 s + fb)][lane][16 B]; lane l supplies frame 16 fb + (l & 15), channels 32 s + 8 (l >> 4) .. + 7.  LDS-DMA with a
    // per-lane source address writes exactly this lane-linear image.
    if constexpr (!FUSED_DW) {
        const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int frag = w * 32 + i;
            const int s = frag >> 3, fb = frag & 7;
            int row = row0 + 16 * fb + lf;
            row = row < M ? row : M - 1;  // rows beyond M are computed on a copy of the last row and never stored
            cx_glds16(y + (long)row * CX_C + 32 * s + 8 * lh, lds0 + frag * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        // Wave w turns out the 32 frames of frame block w: a lane owns channels 4 l .. 4 l + 3 and 256 + 4 l .. + 3 of
        // every row (whole 2 KiB rows per wave load, LayerNorm sums on the DPP path, nothing crosses waves).  Groups of 4
        // frames: their 10 input rows sit in registers, the 4 new rows of the next group are in flight meanwhile.
        // Same arithmetic, in the same order, as swc_dwconv7_ln (bias, taps 0..6 as fma, two-pass LayerNorm).
        const float4* x4 = reinterpret_cast<const float4*>(x);
        // the 7 x 8 taps stay in registers; bias and the LayerNorm affine (used once per group) are re-read from the cache
        // per group so that TWO groups of new rows can be in flight (the rows come from beyond L2: one group ahead left
        // most of their latency exposed, 26 us per block against 21 us for the stand-alone kernel)
        float4 wr[7][2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 7; ++j) wr[j][k] = reinterpret_cast<const float4*>(fr.w7 + (long)j * CX_C)[lane + 64 * k];
        const int f0w = row0 + 32 * w;  // first row of this wave
        // Waves whose 38-row window lies inside one utterance and inside the tensor (all but one in ~8 at T = 1000) skip
        // every boundary test: the tests are integer divisions on the scalar unit, 54 per group
        const bool interior = f0w - 3 >= 0 && f0w + 35 < M && (f0w - 3) / fr.T == (f0w + 35) / fr.T;
        auto load_row = [&](int r, float4 (&dst)[2]) {  // rows outside the tensor are zeros (wave-uniform test)
            const bool ok = interior || (r >= 0 && r < M);
#pragma unroll
            for (int k = 0; k < 2; ++k) dst[k] = ok ? x4[(long)r * (CX_C / 4) + lane + 64 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // A row stays in the window for up to three groups and can be a valid tap for one frame and lie beyond an utterance
        // boundary for another: rows are loaded as they are, the utterance test is made per (frame, row) pair when used.
        float4 win[10][2], nxA[4][2], nxB[4][2];
#pragma unroll
        for (int p = 0; p < 10; ++p) load_row(f0w - 3 + p, win[p]);
#pragma unroll
        for (int p = 0; p < 4; ++p) load_row(f0w + 4 + 3 + p, nxA[p]);  // the new rows of group 1
        const float inv_c = 1.0f / (float)CX_C;
        char* ybase = smem;
        auto group = [&](int g, float4 (&cur)[4][2], float4 (&pre)[4][2]) {
            // `cur`: the 4 new rows of group g + 1 (already in flight); `pre`: where those of group g + 2 are loaded now
            if (g + 2 < 8) {
#pragma unroll
                for (int p = 0; p < 4; ++p) load_row(f0w + 4 * (g + 2) + 3 + p, pre[p]);
            }
            float4 br[2], gw[2], gb[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                br[k] = reinterpret_cast<const float4*>(fr.bias)[lane + 64 * k];
                gw[k] = reinterpret_cast<const float4*>(fr.ln_w)[lane + 64 * k];
                gb[k] = reinterpret_cast<const float4*>(fr.ln_b)[lane + 64 * k];
            }
            int ur[10], uf[4];  // utterance of every window row / frame of the group (boundary waves only)
            if (!interior) {
#pragma unroll
                for (int p = 0; p < 10; ++p) {
                    const int r = f0w + 4 * g - 3 + p;
                    ur[p] = r < 0 ? -1 : r / fr.T;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) uf[u] = ur[u + 3];
            }
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 2; ++k) v[u][k] = br[k];
#pragma unroll
            for (int p = 0; p < 10; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = p - u;  // frame u takes window row p as tap j
                    if (j >= 0 && j < 7) {
                        const bool ok = interior || ur[p] == uf[u];  // same utterance (wave-uniform)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float4 xv = ok ? win[p][k] : make_float4(0.f, 0.f, 0.f, 0.f);
                            v[u][k].x += xv.x * wr[j][k].x; v[u][k].y += xv.y * wr[j][k].y;
                            v[u][k].z += xv.z * wr[j][k].z; v[u][k].w += xv.w * wr[j][k].w;
                        }
                    }
                }
            float sum[4], sq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sum[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) sum[u] += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum[u] = wave_sum_dpp(sum[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float mean = sum[u] * inv_c;
                sq[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    v[u][k].x -= mean; v[u][k].y -= mean; v[u][k].z -= mean; v[u][k].w -= mean;
                    sq[u] += (v[u][k].x * v[u][k].x + v[u][k].y * v[u][k].y) + (v[u][k].z * v[u][k].z + v[u][k].w * v[u][k].w);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sq[u] = wave_sum_dpp(sq[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float rstd = rsqrtf(sq[u] * inv_c + fr.eps);
                const int fl = 4 * g + u;  // frame inside this wave's block
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float o0 = v[u][k].x * rstd * gw[k].x + gb[k].x, o1 = v[u][k].y * rstd * gw[k].y + gb[k].y;
                    const float o2 = v[u][k].z * rstd * gw[k].z + gb[k].z, o3 = v[u][k].w * rstd * gw[k].w + gb[k].w;
                    // channels c = 256 k + 4 l .. + 3 -> fragment (s = c / 32, fb = 2 w + fl / 16), lane' = 16 ((c % 32) / 8) + fl % 16,
                    // byte (c % 8) * 2
                    const int s_ = 8 * k + (lane >> 3);
                    const int off = ((s_ * 8 + 2 * w + (fl >> 4)) * 64 + 16 * ((lane >> 1) & 3) + (fl & 15)) * 16 + (lane & 1) * 8;
                    *reinterpret_cast<uint2*>(ybase + off) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
                }
            }
            // slide the window by 4 rows
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) win[p][k] = win[p + 4][k];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) win[6 + p][k] = cur[p][k];
        };
        for (int g = 0; g < 8; g += 2) {
            group(g, nxA, nxB);
            group(g + 1, nxB, nxA);
        }
        __syncthreads();
    }
    const u32x4* ylds = reinterpret_cast<const u32x4*>(smem) + lane;
    u32x4* hlds = reinterpret_cast<u32x4*>(smem + CX_Y_BYTES) + lane;

    // ---- weight stream of this wave
    // address = wave-uniform byte pointer (SGPR pair, advanced once per phase) + one 32-bit lane offset + immediate:
    // per-fragment 64-bit VGPR addresses cost 28 registers and a spill in the slice loop
    const long per_wave = (long)NS * 64 + CX_PF;  // fragments
    const char* wbase = reinterpret_cast<const char*>(wstream) + (long)w * per_wave * 1024;
    const unsigned lane_off = (unsigned)lane * 16u;
    auto wfrag = [&](int i) -> u32x4 {  // fragment i of the current phase (i may run CX_PF past its end)
        if (CX_ABL & 4) i &= CX_PF - 1;
        return *reinterpret_cast<const u32x4*>(wbase + (long)i * 1024 + lane_off);
    };
    u32x4 ring[CX_PF];
#pragma unroll
    for (int i = 0; i < CX_PF; ++i) ring[i] = wfrag(i);

    f32x4 acc2[8][8];  // [n tile of this wave][frame block]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc1[2][8];  // [hidden tile][frame block]: H^T of this wave's 32 hidden rows

    // Every k-step is its own scheduling region (sched_barrier at its end): the step issues the refill of the ring slot it
    // consumes and the LDS reads of the NEXT step's B fragments, then its MFMAs.  Left to itself hipcc sinks the ring
    // loads to the end of the unrolled trip (issue -> use distance of two MFMAs instead of eight steps) and reads each B
    // fragment right in front of its MFMA (LDS latency exposed on every pair, one wave per SIMD has nobody to hide it).
    // B fragments are read half a k-step (4 frame blocks) ahead: a whole step ahead would keep 16 fragments = 64 registers live
    auto y_frags = [&](int s, int half, u32x4 (&dst)[4]) {
        if ((CX_ABL & 8) && s > 0) return;
#pragma unroll
        for (int b = 0; b < 4; ++b) dst[b] = ylds[(s * 8 + 4 * half + b) * 64];
    };
    auto h_frags = [&](int q, int half, u32x4 (&dst)[4]) {
        if ((CX_ABL & 8) && q > 0) return;
#pragma unroll
        for (int b = 0; b < 4; ++b) dst[b] = hlds[(q * 8 + 4 * half + b) * 64];
    };
    // b1 of this wave's 32 hidden rows, one slice ahead: register r of hidden tile h in a lane of group lh belongs to row
    // 16 h + 4 lh + r.  Loaded a whole slice before its use, so waiting for it never drains the younger ring loads
    float4 bias_nx[2];
    auto load_bias = [&](int j) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            bias_nx[h] = *reinterpret_cast<const float4*>(b1 + (long)j * CX_SL + 32 * w + 16 * h + 4 * lh);
    };
    load_bias(0);
    auto gemm1 = [&](int j) {
        // accumulators start at the bias
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc1[h][b] = (f32x4){bias_nx[h].x, bias_nx[h].y, bias_nx[h].z, bias_nx[h].w};
        load_bias(j + 1 < NS ? j + 1 : 0);
        u32x4 yA[4], yB[4];
        y_frags(0, 0, yA);
        // fully unrolled (a rolled loop makes the ring loop-carried: see the 32 x 32 x 16 kernel)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int a0 = (2 * s) % CX_PF, a1 = (2 * s + 1) % CX_PF;
            {
                y_frags(s, 1, yB);
                mfma16x8_vgpr(ring[a0], ring[a1], yA, acc1[0][0], acc1[0][1], acc1[0][2], acc1[0][3], acc1[1][0], acc1[1][1],
                              acc1[1][2], acc1[1][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                y_frags((s + 1) & 15, 0, yA);  // the last step re-reads step 0 (harmless)
                mfma16x8_vgpr(ring[a0], ring[a1], yB, acc1[0][4], acc1[0][5], acc1[0][6], acc1[0][7], acc1[1][4], acc1[1][5],
                              acc1[1][6], acc1[1][7]);
                ring[a0] = wfrag(2 * s + CX_PF);  // refill the slots just consumed
                ring[a1] = wfrag(2 * s + 1 + CX_PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // MFMA results in VGPRs -> VALU readers: the wait states hipcc would insert for its own MFMAs (asm is opaque to it)
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc1[0][0]), "+v"(acc1[0][1]), "+v"(acc1[0][2]), "+v"(acc1[0][3]), "+v"(acc1[0][4]), "+v"(acc1[0][5]),
                       "+v"(acc1[0][6]), "+v"(acc1[0][7]), "+v"(acc1[1][0]), "+v"(acc1[1][1]), "+v"(acc1[1][2]), "+v"(acc1[1][3]),
                       "+v"(acc1[1][4]), "+v"(acc1[1][5]), "+v"(acc1[1][6]), "+v"(acc1[1][7]));
        wbase += 32 * 1024;
    };
    // GELU of the two hidden tiles of frame block b -> one packed B fragment of GEMM2 (this wave's k-step), kept IN PLACE of
    // hidden tile 0: the packed copy of the slice costs no registers of its own
    auto gelu_frag = [&](int b) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (CX_ABL & 2) ? acc1[e >> 2][b][e & 3] : gelu_fast(acc1[e >> 2][b][e & 3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[0][b][i] = __uint_as_float(pack_bf16x2(v[2 * i], v[2 * i + 1]));
    };
    auto store_h = [&]() {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            hlds[(w * 8 + b) * 64] = (u32x4){__float_as_uint(acc1[0][b][0]), __float_as_uint(acc1[0][b][1]),
                                             __float_as_uint(acc1[0][b][2]), __float_as_uint(acc1[0][b][3])};
    };
    // GEMM2 over the slice whose H^T is in LDS; `with_gelu`: the GELU of the NEXT slice (acc1) rides along, two packed
    // fragments per k-step, in the same scheduling region as that step's 64 MFMAs
    auto gemm2 = [&](auto with_gelu) {
        u32x4 hA[4], hB[4];
        h_frags(0, 0, hA);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            {
                h_frags(q, 1, hB);
#pragma unroll
                for (int n = 0; n < 8; ++n) mfma16x4_agpr(ring[n], hA, acc2[n][0], acc2[n][1], acc2[n][2], acc2[n][3]);
                if constexpr (decltype(with_gelu)::value) gelu_frag(2 * q);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                if (q + 1 < 4) h_frags(q + 1, 0, hA);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    mfma16x4_agpr(ring[n], hB, acc2[n][4], acc2[n][5], acc2[n][6], acc2[n][7]);
                    ring[n] = wfrag(q * 8 + n + CX_PF);
                }
                if constexpr (decltype(with_gelu)::value) gelu_frag(2 * q + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wbase += 32 * 1024;
    };

    // ---- pipeline over the hidden slices
    gemm1(0);
#pragma unroll
    for (int b = 0; b < 8; ++b) gelu_frag(b);
    store_h();
    __syncthreads();
    for (int j = 1; j < NS; ++j) {
        gemm1(j);
        gemm2(std::true_type{});
        if (!(CX_ABL & 1)) __syncthreads();  // every wave has read H_{j-1}
        store_h();
        if (!(CX_ABL & 1)) __syncthreads();  // H_j visible
    }
    gemm2(std::false_type{});
    __syncthreads();  // LDS is free: the epilogue re-uses all of it
    // (the MFMAs above are asm: the wait states between the last of them and the first v_accvgpr_read of the epilogue are ours;
    // the barrier alone already takes longer, this documents the requirement)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    if (CX_ABL & 16) {  // keep the accumulators alive, store nothing
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) keep += acc2[a][b][0] + acc2[a][b][3];
        if (keep == 12345.678f) xo[0] = keep;
        return;
    }
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
#if CX_EPI_PREFETCH
        // the residual rows of this pass first: 32 independent 16-byte loads per lane, in flight across the LDS round trip
        // (the arithmetic registers of the slice loop are free here)
        float4 rr[16][2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long row = (long)row0 + 64 * p + 16 * w + i;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                rr[i][hf] = row < M ? *reinterpret_cast<const float4*>(x + row * CX_C + 256 * hf + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#endif
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int fbl = 0; fbl < 4; ++fbl) {
                const f32x4& t = acc2[n][4 * p + fbl];  // channels 128 w + 16 n + 4 lh + r of frame 16 fbl + lf of this pass
                *reinterpret_cast<float4*>(tl + (16 * fbl + lf) * CX_TLD + 128 * w + 16 * n + 4 * lh) = make_float4(t[0], t[1], t[2], t[3]);
            }
        __syncthreads();
#if CX_EPI_PREFETCH
#pragma unroll
#else
#pragma unroll 4
#endif
        for (int i = 0; i < 16; ++i) {
            const int fl = 16 * w + i;
            const long row = (long)row0 + 64 * p + fl;
            if (row < M) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 v = *reinterpret_cast<const float4*>(tl + fl * CX_TLD + 256 * hf + 4 * lane);
#if CX_EPI_PREFETCH
                    float4 r = rr[i][hf];
#else
                    float4 r = *reinterpret_cast<const float4*>(x + row * CX_C + 256 * hf + 4 * lane);
#endif
                    r.x += g4[hf].x * (v.x + c4[hf].x); r.y += g4[hf].y * (v.y + c4[hf].y);
                    r.z += g4[hf].z * (v.z + c4[hf].z); r.w += g4[hf].w * (v.w + c4[hf].w);
                    *reinterpret_cast<float4*>(xo + row * CX_C + 256 * hf + 4 * lane) = r;
                }
            }
        }
        __syncthreads();
    }
}

// the packed stream for convnext16_kernel: the same phases as convnext_pack_kernel.  G1(j): fragment i = k-step i / 2, hidden
// tile i % 2; G2(j): fragment i = k-step i / 8 (producing wave), n tile i % 8, hidden values in the order the in-register
// conversion of GEMM1's accumulators produces them
__global__ void convnext16_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, uint4* __restrict__ out,
                                       int NS) {
    const long per_wave = (long)NS * 64 + CX_PF;
    const long total = 4 * per_wave * 64;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int lane = (int)(id & 63);
    const long f_all = id >> 6;
    const int w = (int)(f_all / per_wave);
    const long f = f_all - (long)w * per_wave;
    const int lf = lane & 15, lh = lane >> 4;
    const long I = (long)NS * CX_SL;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (f < (long)NS * 64) {
        const int p = (int)(f >> 5), i = (int)(f & 31);
        const bool is_g1 = p == 0 || ((p & 1) && p < 2 * NS - 1);
        if (is_g1) {
            const int j = p == 0 ? 0 : (p + 1) >> 1;
            const int s = i >> 1, h = i & 1;
            const long row = (long)j * CX_SL + 32 * w + 16 * h + lf;  // hidden row
            v = *reinterpret_cast<const uint4*>(w1 + row * CX_C + 32 * s + 8 * lh);
        } else {
            const int j = p == 2 * NS - 1 ? NS - 1 : (p >> 1) - 1;
            const int q = i >> 3, n = i & 7;
            const long nrow = 128 * w + 16 * n + lf;  // output column of the block = row of W2
            const long hid = (long)j * CX_SL + 32 * q + 4 * lh;  // elements jj: hid + 16 (jj >> 2) + (jj & 3)
            const uint2 lo = *reinterpret_cast<const uint2*>(w2 + nrow * I + hid);
            const uint2 hi = *reinterpret_cast<const uint2*>(w2 + nrow * I + hid + 16);
            v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    out[id] = v;
}

#endif  // CX_MFMA16

// One thread per 16-byte chunk of the packed stream.  Stream of wave w: for slice j: [GEMM1(j) fragments: W1 rows
// 128 j + 32 w .. + 31, k-steps s = 0..31], and after GEMM1(j) for j >= 1 (and once more at the end) the GEMM2 fragments
// of slice j - 1: for k-step q = 0..7, n block 4 w + n, n = 0..3.  Consumption order: G1(0), G1(1), G2(0), G1(2), G2(1), ...
// gamma (optional): W2 rows are scaled by it before the bf16 rounding (CX_RES_ACC: the kernel then never multiplies by gamma)
__global__ void convnext_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, uint4* __restrict__ out,
                                     int NS, const float* __restrict__ gamma) {
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
            if (gamma) {
                const float gsc = gamma[nrow];
                unsigned* pv = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = bf16_to_f32((bf16_t)(pv[k] & 0xffffu)) * gsc, b = bf16_to_f32((bf16_t)(pv[k] >> 16)) * gsc;
                    pv[k] = (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
                }
            }
        }
    }
    out[id] = v;
}

}  // namespace

extern "C" int64_t swc_convnext_stream_bytes(int32_t C, int32_t I) {
    if (C != CX_C || I <= 0 || I % CX_SL != 0) return 0;
    return 4L * ((long)(I / CX_SL) * 64 + CX_PF) * 1024;
}

extern "C" int swc_convnext_pack(const void* w1, const void* w2, const float* gamma, void* stream_out, int32_t C, int32_t I,
                                 void* stream) {
    SWC_CHECK_ARG(w1 && w2 && gamma && stream_out, "swc_convnext_pack: null pointer");
    SWC_CHECK_ARG(C == CX_C && I > 0 && I % CX_SL == 0, "swc_convnext_pack: needs C = %d and I a multiple of %d (C=%d I=%d)",
                  CX_C, CX_SL, C, I);
    SWC_CHECK_ARG(aligned16(w1) && aligned16(w2) && aligned16(stream_out), "swc_convnext_pack: unaligned");
    const int NS = I / CX_SL;
    const long total = 4L * ((long)NS * 64 + CX_PF) * 64;
#if CX_MFMA16
    hipLaunchKernelGGL(convnext16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w1, (const bf16_t*)w2, (uint4*)stream_out, NS);
#else
    hipLaunchKernelGGL(convnext_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w1, (const bf16_t*)w2, (uint4*)stream_out, NS, CX_RES_ACC ? gamma : (const float*)nullptr);
#endif
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
#if CX_MFMA16
    auto kern = convnext16_kernel<false>;
#else
    auto kern = convnext_mlp_kernel<false>;
#endif
    SWC_ENABLE_LDS(kern, CX_LDS, "swc_convnext_mlp");
    const unsigned grid = (unsigned)((M + CX_BM - 1) / CX_BM);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), CX_LDS, (hipStream_t)stream, (const bf16_t*)y, (const u32x4*)w_stream,
                       b1, b2, gamma, x, x, M, I / CX_SL, CxFront{});
    SWC_CHECK_LAUNCH("swc_convnext_mlp");
    return SWC_OK;
}

extern "C" int swc_convnext_block(const float* x, float* x_out, const float* dw_w7, const float* dw_bias, const float* ln_w,
                                  const float* ln_b, float eps, const void* w_stream, const float* b1, const float* b2,
                                  const float* gamma, int32_t B, int32_t T, int32_t C, int32_t I, const int32_t* t_limit,
                                  int32_t operand_dtype, void* stream) {
    SWC_CHECK_ARG(operand_dtype == SWC_BF16 || operand_dtype == SWC_F16, "swc_convnext_block: operand_dtype must be BF16 or F16");
#if CX_MFMA16 || CX_RES_ACC
    SWC_CHECK_ARG(operand_dtype == SWC_BF16, "swc_convnext_block: this build has bf16 operands only");
#endif
    SWC_CHECK_ARG(x && x_out && x != x_out, "swc_convnext_block: x and x_out must be two different buffers");
    SWC_CHECK_ARG(x && dw_w7 && dw_bias && ln_w && ln_b && w_stream && b1 && b2 && gamma, "swc_convnext_block: null pointer");
    SWC_CHECK_ARG(C == CX_C && I > 0 && I % CX_SL == 0, "swc_convnext_block: needs C = %d and I a multiple of %d (C=%d I=%d)",
                  CX_C, CX_SL, C, I);
    SWC_CHECK_ARG(B >= 0 && T >= 0 && (long)B * T < (1L << 31), "swc_convnext_block: bad B/T");
    SWC_CHECK_ARG(aligned16(x) && aligned16(x_out) && aligned16(dw_w7) && aligned16(dw_bias) && aligned16(ln_w) && aligned16(ln_b) &&
                      aligned16(w_stream) && aligned16(b1) && aligned16(b2) && aligned16(gamma),
                  "swc_convnext_block: unaligned");
    const int M = B * T;
    if (M == 0) return SWC_OK;
    const unsigned grid = (unsigned)((M + CX_BM - 1) / CX_BM);
    CxFront fr{dw_w7, dw_bias, ln_w, ln_b, T, eps, t_limit};
#if CX_MFMA16
    auto kern = convnext16_kernel<true>;
#else
    if (operand_dtype == SWC_F16) {
        auto kern16 = convnext_mlp_kernel<true, true>;
        SWC_ENABLE_LDS(kern16, CX_LDS, "swc_convnext_block");
        hipLaunchKernelGGL(kern16, dim3(grid), dim3(256), CX_LDS, (hipStream_t)stream, (const bf16_t*)nullptr,
                           (const u32x4*)w_stream, b1, b2, gamma, x, x_out, M, I / CX_SL, fr);
        SWC_CHECK_LAUNCH("swc_convnext_block");
        return SWC_OK;
    }
    auto kern = convnext_mlp_kernel<true>;
#endif
    SWC_ENABLE_LDS(kern, CX_LDS, "swc_convnext_block");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), CX_LDS, (hipStream_t)stream, (const bf16_t*)nullptr,
                       (const u32x4*)w_stream, b1, b2, gamma, x, x_out, M, I / CX_SL, fr);
    SWC_CHECK_LAUNCH("swc_convnext_block");
    return SWC_OK;
}
