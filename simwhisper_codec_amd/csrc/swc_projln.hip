// swc_proj_ln: a split-f16 projection onto the residual stream WITH the LayerNorm behind it, in one kernel (modules.py:214-232:
// `x = x + out_proj(attn)` -> `self_attn_layer_norm`-style normalisation of the next sub-block, and `x = x + fc2(h)` -> the next
// layer's first LayerNorm):
//     x_out[M][768]  = x + alpha * (A W^T) + bias                                  residual stream, f32
//     y_next[M][768] = LayerNorm(x_out; ln_w, ln_b) as split-f16 at SWC_F16S_ACT_SCALE   (optional) the next GEMM's operand
// A [M][K] and W [768][K] are split-f16 (three f16 MFMAs per product: hi*hi + hi*lo + lo*hi, f32 accumulate: f32-class results).
// The two-launch form writes the residual row (f32), reads it back in swc_layernorm and writes the operand; here the row is
// normalised by the wave that stores it, while it is still in registers: one launch and 49 MB of HBM reads less per call at
// 16 000 tokens, and no `layernorm2_kernel` launch in the encoder's layers.
// MEASURED (round 4, profiles/r04_proj_ln_split_f16.txt): 85 against 78 us (K = 768) and 222 against 194 us (K = 3072) for swc_gemm +
// swc_layernorm at 16 000 tokens — the full-row tile streams all of W through every CU (1.8 x the L2 -> CU traffic of swc_gemm's
// 192 x 256 tiles) and the row swc_layernorm reads back comes out of the Infinity Cache anyway.  The codec does NOT call this kernel;
// it stays in the library, parity-tested, as the measured answer to "fold LayerNorm into the f32-output GEMM epilogue".
//
// Geometry (the full-row tile of swc_layer_tail, csrc/swc_mlp.hip): a workgroup owns 64 tokens x all 768 columns, 4 waves, one per
// SIMD; wave w accumulates out^T[192 columns of its own][64 tokens] in 192 registers.  The tokens are the MFMA's B operand: their
// tile of A goes through LDS by LDS-DMA in stages of 64 k (16 fragments of 1 KiB per stage, 4 per wave, 4 stages in the ring, one
// workgroup barrier per stage = per 144 MFMAs of a wave).  WEIGHTS NEVER TOUCH LDS: every weight fragment has one consumer wave and
// goes global -> VGPR as a contiguous 1 KiB wave load from a stream swc_proj_ln_pack lays out in each wave's order of consumption
// (per k-step of 16: the 6 lo fragments, then the 6 hi fragments of its 6 column blocks), PL_PF fragments in flight per wave.
// Epilogue: out^T through LDS (transposed, two passes of 32 tokens) so that the residual stream is read and written in whole 3 KiB
// rows; element arithmetic and order are those of swc_gemm's epilogue (acc * alpha + bias, + residual) and of swc_layernorm (row
// sums on the DPP path, two-pass variance), so LayerNorm(x_out) is bit-identical to swc_layernorm on the row this kernel stored.
//
// MFMA: v_mfma_f32_32x32x16_f16.  Operand maps (lane l): A[row l&31][k = 8(l>>5) + j], B[k = 8(l>>5) + j][col l&31],
// D[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31], r = 0..15.
#include "swc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PL_N = 768;          // output columns = width of the residual stream
constexpr int PL_BM = 64;          // tokens per workgroup
constexpr int PL_NB = PL_N / 4 / 32;  // 32-column blocks of out^T per wave (6)
constexpr int PL_KST = 64;         // k per stage of the A ring
constexpr int PL_STAGES = 4;       // stages in the ring (16 KiB each)
constexpr int PL_STAGE_BYTES = PL_BM * PL_KST * 4;  // 64 tokens x 64 k x (hi + lo)
#ifndef PL_PF
#define PL_PF 24                   // weight fragments in flight per wave (1 KiB each): two k-steps
#endif
constexpr int PL_FPS = (PL_KST / 16) * 2 * PL_NB;   // weight fragments per stage and wave (48)
static_assert(PL_FPS % PL_PF == 0, "the ring index of a fragment must not depend on the stage");
constexpr int PL_TLD = PL_N + 4;   // row pitch (floats) of the epilogue transpose buffer: conflict-free b128 writes
constexpr int PL_LDS = 32 * PL_TLD * 4 > PL_STAGES * PL_STAGE_BYTES ? 32 * PL_TLD * 4 : PL_STAGES * PL_STAGE_BYTES;
// Timing ablations for tuning builds only (-DPL_ABL=mask, wrong results): 1 = no barriers in the stage loop, 4 = no weight loads
// in the loop (the ring keeps its first fragments), 8 = no A staging after the first stages, 16 = no epilogue
#ifndef PL_ABL
#define PL_ABL 0
#endif

struct PlNorm {
    const float* w;  // [768]
    const float* b;  // [768]
};

__device__ __forceinline__ f32x16 pl_mfma(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
}

__device__ __forceinline__ void pl_glds16(const void* gsrc, unsigned lds_addr) {
    // one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS at lds_addr (wave-uniform)
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}

// wstream: per wave w (4 of them) (K / 16) * 12 + PL_PF fragments of 1 KiB in the order of consumption (projln_pack_kernel)
__global__ __launch_bounds__(256, 1) void projln_kernel(const char* __restrict__ a, long a_pitch /* bytes per row of A */,
                                                        const u32x4* __restrict__ wstream, const float* __restrict__ bias,
                                                        float alpha, const float* x, float* xo, PlNorm nln, float eps,
                                                        unsigned short* __restrict__ y_next, int M, int K, unsigned* sat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lf = lane & 31, lh = lane >> 5;
    const int row0 = blockIdx.x * PL_BM;
    const int NST = K / PL_KST;

    const long per_wave = (long)(K / 16) * 2 * PL_NB + PL_PF;  // fragments
    const char* wbase = reinterpret_cast<const char*>(wstream) + (long)w * per_wave * 1024;
    const unsigned lane_off = (unsigned)lane * 16u;
    auto wfrag = [&](int i) __attribute__((always_inline)) -> u32x4 {  // fragment i of the current stage (i may run PL_PF past its end)
        if (PL_ABL & 4) i &= PL_PF - 1;
        return *reinterpret_cast<const u32x4*>(wbase + (long)i * 1024 + lane_off);
    };
    u32x4 ring[PL_PF];

    // ---- A staging: stage t = k 64 t .. 64 t + 63 as 16 fragments [k-step ks (4)][plane: hi, lo][token block fb (2)] of 1 KiB.
    // Wave w issues the 4 fragments of k-step ks = w: lane l supplies token 32 fb + (l & 31), k = 16 ks + 8 (l >> 5) .. + 7 — in the
    // split row that is 16 bytes at [32-block (2 t + ks / 2)] * 128 + plane * 64 + (ks & 1) * 32 + (l >> 5) * 16
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const char* arow[2];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        int row = row0 + 32 * fb + lf;
        row = row < M ? row : M - 1;  // rows beyond M are computed on a copy of the last row and never stored
        arow[fb] = a + (long)row * a_pitch + (w >> 1) * 128 + (w & 1) * 32 + lh * 16;
    }
    auto stage_dma = [&](int t) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (t & (PL_STAGES - 1)) * PL_STAGE_BYTES + w * 4096;
#pragma unroll
        for (int plane = 0; plane < 2; ++plane)
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) pl_glds16(arow[fb] + (long)t * 256 + plane * 64, dst + (plane * 2 + fb) * 1024);
    };
    const u32x4* alds = reinterpret_cast<const u32x4*>(smem) + lane;
    auto a_frags = [&](int t, int ks, u32x4 (&xh)[2], u32x4 (&xl)[2]) __attribute__((always_inline)) {
        const int base = ((t & (PL_STAGES - 1)) * PL_STAGE_BYTES + ks * 4096) / 16;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            xh[fb] = alds[base + fb * 64];
            xl[fb] = alds[base + (2 + fb) * 64];
        }
    };

    f32x16 acc[PL_NB][2];  // [column block of this wave][token block]: out^T, 192 accumulators (AGPRs)
#pragma unroll
    for (int n = 0; n < PL_NB; ++n)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][b][r] = 0.f;

    stage_dma(0);
    if (NST > 1) stage_dma(1);
    // in slot order, pinned: hipcc's s_waitcnt bookkeeping merges the age of every ring register at the loop header from both
    // entries; with the preload in any other order than the loop's own refills it drained the whole ring once per stage
#pragma unroll
    for (int i = 0; i < PL_PF; ++i) {
        ring[i] = wfrag(i);
        __builtin_amdgcn_sched_barrier(0);
    }

    // One k-step = its own scheduling region: the LDS reads of the NEXT k-step's token fragments, 36 MFMAs, the refill of the 12
    // ring slots it consumed.  Order inside: the lo weight fragments first (their slots are refilled earliest), every accumulator
    // is touched once per 12 MFMAs
    auto kstep = [&](int i0, const u32x4 (&xh)[2], const u32x4 (&xl)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < PL_NB; ++n) {
            const int sl = (i0 + n) % PL_PF;
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[n][b] = pl_mfma(ring[sl], xh[b], acc[n][b]);  // W_lo . x_hi
            ring[sl] = wfrag(i0 + n + PL_PF);
        }
#pragma unroll
        for (int n = 0; n < PL_NB; ++n) {
            const int sl = (i0 + PL_NB + n) % PL_PF;
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[n][b] = pl_mfma(ring[sl], xl[b], acc[n][b]);  // W_hi . x_lo
        }
#pragma unroll
        for (int n = 0; n < PL_NB; ++n) {
            const int sl = (i0 + PL_NB + n) % PL_PF;
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[n][b] = pl_mfma(ring[sl], xh[b], acc[n][b]);  // W_hi . x_hi
            ring[sl] = wfrag(i0 + PL_NB + n + PL_PF);
        }
    };

    for (int t = 0; t < NST; ++t) {
        // stage t's fragments were requested two stages (or, in front of the loop, PL_PF weight loads) ago: everything but the
        // youngest PL_PF loads of this wave has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL_PF) : "memory");
        if (!(PL_ABL & 1)) __syncthreads();  // ... and the other waves' too; every wave has finished reading stage t - 1
        if (t + 2 < NST && !((PL_ABL & 8) && t > 1)) stage_dma(t + 2);
        u32x4 hA[2], lA[2], hB[2], lB[2];
        a_frags(t, 0, hA, lA);
        __builtin_amdgcn_sched_barrier(0);
        a_frags(t, 1, hB, lB);
        kstep(0, hA, lA);
        __builtin_amdgcn_sched_barrier(0);
        a_frags(t, 2, hA, lA);
        kstep(12, hB, lB);
        __builtin_amdgcn_sched_barrier(0);
        a_frags(t, 3, hB, lB);
        kstep(24, hA, lA);
        __builtin_amdgcn_sched_barrier(0);
        kstep(36, hB, lB);
        __builtin_amdgcn_sched_barrier(0);
        wbase += PL_FPS * 1024;
    }
    __syncthreads();  // LDS is free: the epilogue re-uses it

    if (PL_ABL & 16) {  // keep the accumulators alive, store nothing
        float keep = 0.f;
#pragma unroll
        for (int n = 0; n < PL_NB; ++n)
#pragma unroll
            for (int b = 0; b < 2; ++b) keep += acc[n][b][0] + acc[n][b][7];
        if (keep == 12345.678f) xo[0] = keep;
        return;
    }
    // ---- epilogue: x_out[row][n] = acc * alpha + bias[n] + x[row][n] via a transposed f32 image [32 tokens][772]; the wave that
    // owns a row then holds all 768 values of it: LayerNorm on the spot
    float* tl = reinterpret_cast<float*>(smem);
    float4 c4[3], nw[3], nb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c4[k] = bias ? *reinterpret_cast<const float4*>(bias + 256 * k + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (y_next) {
            nw[k] = *reinterpret_cast<const float4*>(nln.w + 256 * k + 4 * lane);
            nb[k] = *reinterpret_cast<const float4*>(nln.b + 256 * k + 4 * lane);
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        // the residual rows of this pass first: 24 independent 16-byte loads per lane, in flight across the LDS round trip
        float4 rr[8][3];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long row = (long)row0 + 32 * p + 8 * w + i;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                rr[i][k] = row < M ? *reinterpret_cast<const float4*>(x + row * PL_N + 256 * k + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int n = 0; n < PL_NB; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& t = acc[n][p];
                *reinterpret_cast<float4*>(tl + lf * PL_TLD + 32 * (PL_NB * w + n) + 8 * g + 4 * lh) =
                    make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            }
        __syncthreads();
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += 4) {
            float4 o[4][3];
            float s[4], q[4], mean[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int fl = 8 * w + i0 + u;
                const long row = (long)row0 + 32 * p + fl;
                s[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(tl + fl * PL_TLD + 256 * k + 4 * lane);
                    float4 r;
                    r.x = v.x * alpha + c4[k].x; r.y = v.y * alpha + c4[k].y; r.z = v.z * alpha + c4[k].z; r.w = v.w * alpha + c4[k].w;
                    r.x += rr[i0 + u][k].x; r.y += rr[i0 + u][k].y; r.z += rr[i0 + u][k].z; r.w += rr[i0 + u][k].w;
                    o[u][k] = r;
                    if (row < M) *reinterpret_cast<float4*>(xo + row * PL_N + 256 * k + 4 * lane) = r;
                    s[u] += (r.x + r.y) + (r.z + r.w);
                }
            }
            if (y_next) {
#pragma unroll
                for (int u = 0; u < 4; ++u) mean[u] = wave_sum_dpp(s[u]) / (float)PL_N;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    q[u] = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float a_ = o[u][k].x - mean[u], b_ = o[u][k].y - mean[u], c = o[u][k].z - mean[u], d = o[u][k].w - mean[u];
                        q[u] += (a_ * a_ + b_ * b_) + (c * c + d * d);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q[u]) / (float)PL_N + eps);
                    const long row = (long)row0 + 32 * p + 8 * w + i0 + u;
                    if (row < M) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float4 t = o[u][k];
                            f16s_store4(y_next + row * (2 * PL_N), 256 * k + 4 * lane,
                                        ((t.x - mean[u]) * rstd * nw[k].x + nb[k].x) * SWC_F16S_ACT_SCALE,
                                        ((t.y - mean[u]) * rstd * nw[k].y + nb[k].y) * SWC_F16S_ACT_SCALE,
                                        ((t.z - mean[u]) * rstd * nw[k].z + nb[k].z) * SWC_F16S_ACT_SCALE,
                                        ((t.w - mean[u]) * rstd * nw[k].w + nb[k].w) * SWC_F16S_ACT_SCALE, amax);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (y_next) sat_commit(sat, 0, amax, SWC_F16S_LIMIT);
}

// One thread per 16-byte chunk of the packed stream.  Stream of wave w: for every k-step s (16 k): fragments 0..5 = the lo halves,
// 6..11 = the hi halves of W rows 192 w + 32 n + (lane & 31) (n = 0..5), k = 16 s + 8 (lane >> 5) .. + 7; then PL_PF zero
// fragments (read ahead, never used).
__global__ void projln_pack_kernel(const unsigned short* __restrict__ wsp, uint4* __restrict__ out, int K) {
    const long per_wave = (long)(K / 16) * 2 * PL_NB + PL_PF;
    const long total = 4 * per_wave * 64;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int lane = (int)(id & 63);
    const long f_all = id >> 6;
    const int w = (int)(f_all / per_wave);
    const long f = f_all - (long)w * per_wave;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (f < (long)(K / 16) * 2 * PL_NB) {
        const int s = (int)(f / (2 * PL_NB)), j = (int)(f % (2 * PL_NB));
        const int n = j % PL_NB, plane = j < PL_NB ? 1 : 0;  // plane 1 = lo
        const long row = 192 * w + 32 * n + (lane & 31);
        v = *reinterpret_cast<const uint4*>(wsp + row * (2L * K) + f16s_col(16 * s + 8 * (lane >> 5)) + 32 * plane);
    }
    out[id] = v;
}

}  // namespace

extern "C" int64_t swc_proj_ln_stream_bytes(int32_t N, int32_t K) {
    if (N != PL_N || K <= 0 || K % PL_KST != 0) return 0;
    return 4L * ((long)(K / 16) * 2 * PL_NB + PL_PF) * 1024;
}

extern "C" int swc_proj_ln_pack(const void* w_f16s, void* stream_out, int32_t N, int32_t K, void* stream) {
    SWC_CHECK_ARG(w_f16s && stream_out, "swc_proj_ln_pack: null pointer");
    SWC_CHECK_ARG(N == PL_N && K > 0 && K % PL_KST == 0, "swc_proj_ln_pack: needs N = %d and K a multiple of %d (N=%d K=%d)", PL_N,
                  PL_KST, N, K);
    SWC_CHECK_ARG(aligned16(w_f16s) && aligned16(stream_out), "swc_proj_ln_pack: unaligned");
    const long total = 4L * ((long)(K / 16) * 2 * PL_NB + PL_PF) * 64;
    hipLaunchKernelGGL(projln_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_f16s, (uint4*)stream_out, K);
    SWC_CHECK_LAUNCH("swc_proj_ln_pack");
    return SWC_OK;
}

extern "C" int swc_proj_ln(const void* a_f16s, int64_t lda, const void* w_stream, const float* bias, float alpha, const float* x,
                           float* x_out, const float* ln_w, const float* ln_b, float eps, void* y_next, int32_t M, int32_t N,
                           int32_t K, void* stream) {
    SWC_CHECK_ARG(a_f16s && w_stream && x && x_out, "swc_proj_ln: null pointer");
    SWC_CHECK_ARG(!y_next || (ln_w && ln_b), "swc_proj_ln: y_next needs ln_w / ln_b");
    SWC_CHECK_ARG(N == PL_N && K > 0 && K % PL_KST == 0, "swc_proj_ln: needs N = %d and K a multiple of %d (N=%d K=%d)", PL_N, PL_KST,
                  N, K);
    SWC_CHECK_ARG(M >= 0 && lda >= K && lda % 4 == 0, "swc_proj_ln: bad M / lda");
    SWC_CHECK_ARG(aligned16(a_f16s) && aligned16(w_stream) && aligned16(bias) && aligned16(x) && aligned16(x_out) && aligned16(ln_w) &&
                      aligned16(ln_b) && aligned16(y_next),
                  "swc_proj_ln: unaligned");
    if (M == 0) return SWC_OK;
    SWC_ENABLE_LDS(projln_kernel, PL_LDS, "swc_proj_ln");
    const unsigned grid = (unsigned)((M + PL_BM - 1) / PL_BM);
    hipLaunchKernelGGL(projln_kernel, dim3(grid), dim3(256), PL_LDS, (hipStream_t)stream, (const char*)a_f16s, (long)lda * 4,
                       (const u32x4*)w_stream, bias, alpha, x, x_out, PlNorm{ln_w, ln_b}, eps, (unsigned short*)y_next, M, K,
                       swc_sat_counter());
    SWC_CHECK_LAUNCH("swc_proj_ln");
    return SWC_OK;
}
