// swc_attention (16-bit operand paths): varlen, non-causal, head_dim 64 flash attention on the
// 16x16x32 MFMAs.
//   PLANES = 1: bf16 q/k/v, v_mfma_f32_16x16x32_bf16, bf16 output            (decode side)
//   PLANES = 2: split-f16 q/k/v (SWC_F16S, scale 64), three f16 MFMAs per product
//               (hi*hi + hi*lo + lo*hi, f32 accumulate): f32-class scores and outputs at 3/16
//               of the exact-f32 MFMA cycles; split-f16 output                (encode side)
// One workgroup = 128 queries of one (utterance, head); each of its 4 waves owns 32 queries (two
// 16-query MFMA tiles that share every K/V fragment) and walks the valid keys in tiles of 64,
// K/V tiles double-buffered in LDS by LDS-DMA.
//
// As in the f32 kernel the score tile is computed transposed (S^T = K Q^T): its C/D fragment is
// already the B operand of O^T = V^T P^T once two 16-key tiles are packed into one 32-deep k-step
// (k index 8h+j  <->  key 16 t0 + 4h + j for j < 4, 16 t1 + 4h + j - 4 otherwise).  The matching
// V^T operand (4 consecutive keys of one channel) comes out of the row-major V tile through
// ds_read_b64_tr_b16, the hardware transposing read.
#include "swc_common.h"
#include <type_traits>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16a[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16a(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}

// max over the lane pair (l, l ^ W), W = 16 or 32, in both lanes: operand rows swapped against a copy of themselves
typedef unsigned u32x2_sw __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ float xor_max(float v) {
    const unsigned u = __float_as_uint(v);
    const u32x2_sw r = W == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false) : __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int PLANES>
__device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 c) {
#if defined(ATT_ABL) && (ATT_ABL & 8)   // timing ablation only (wrong results): no matrix instruction, operands kept alive
    c[0] += __uint_as_float(a.x ^ b.x);
    return c;
#endif
    if constexpr (PLANES == 1)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                       *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a),
                                                      *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
}

// two f32 -> one dword of two 16-bit values (RNE) in ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32): as a vector
// conversion, not inline asm — the operands come straight from v_exp_f32, and only for instructions it knows does hipcc
// insert the wait state a transcendental result needs before its first use (an asm form returned NaNs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2n __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    const bf16x2n r = __builtin_convertvector((f32x2){lo, hi}, bf16x2n);
    return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ unsigned pack2_f16(float lo, float hi) {
    const f16x2 r = __builtin_convertvector((f32x2){lo, hi}, f16x2);
    return *reinterpret_cast<const unsigned*>(&r);
}


// KT16: keys per tile (64, or 128 for the bf16 kernel: half the fences and dependent softmax chains per key)
// NQT: 16-query MFMA tiles per wave (2: 128 queries per workgroup; 1: 64 — half the registers, twice the waves per SIMD)
// RES (build option -DATT_RES=1, bf16 only, T <= 512; round 4, measured, not the default): one workgroup of 8 waves per
// (utterance, head) keeps ALL of its K and V in LDS (8 tiles x 16 KiB = 128 KiB), staged once, and its waves walk the query
// blocks (32 queries per wave, 256 per pass) with no further staging, fence or barrier.  Removes the re-staging of K / V by every
// 128-query workgroup (4 x per (utterance, head) at T = 500) at the price of one workgroup per CU and 384 workgroups on 256 CUs.
template <int PLANES, int KT16, int NQT, bool RES = false>
__global__ __launch_bounds__(RES ? 512 : 256, RES ? 1 : (NQT == 1 ? 4 : 2)) void attn16_kernel(const char* __restrict__ qkv, unsigned short* __restrict__ out,
                                                        const int* __restrict__ lens, int T, int H,
                                                        const int* __restrict__ row_start) {
    constexpr int ROWB = 128 * PLANES;    // bytes of one (token, head) row: 64 bf16, or [32 hi|32 lo|32 hi|32 lo] f16
    constexpr int CPR = ROWB / 16;        // 16-byte chunks per row
    constexpr int TILE = KT16 * ROWB;     // bytes of a K or V tile
    constexpr int QB16 = 64 * NQT;        // queries per workgroup
    constexpr int QW = 16 * NQT;          // queries per wave
    constexpr int NKS = KT16 / 16;        // 16-key score sub-tiles per tile
    constexpr int NPR = KT16 / 32;        // 32-key k-steps of the PV product per tile
    constexpr int NI = TILE / 1024;       // LDS-DMA wave-instructions per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]

    constexpr int NW = RES ? 8 : 4;       // waves per workgroup
    const int b = blockIdx.z, head = blockIdx.y;
    int q0 = RES ? 0 : blockIdx.x * QB16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fh = lane >> 4;
    const int D = H * 64;
    const long ldb = 3L * D * 2 * PLANES;  // bytes per token row of qkv
    const long part = (long)D * 2 * PLANES;  // byte offset between the q, k and v parts
    int len = lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    // padded layout: utterance b owns rows b * T .. b * T + T - 1; packed (row_start != NULL): rows row_start[b] ..
    // row_start[b] + len - 1 and nothing else — rows beyond its length belong to the next utterance and are neither read
    // nor written
    const long r0 = row_start ? (long)row_start[b] : (long)b * T;
    const int qlim = row_start ? len : T;
    const char* base = qkv + r0 * ldb + (long)head * ROWB;

    auto krow_swz = [](int row) { return PLANES == 1 ? (((row >> 1) ^ ((row >> 4) << 1)) & 7) : (row & 15); };
    auto vrow_swz = [](int row) { return PLANES == 1 ? ((row >> 1) & 3) : (row & 7); };

    // ---- output addressing
    auto store_o = [&](int q, int d, float a, float bq, float c, float e) {
        if (q >= qlim) return;
        if constexpr (PLANES == 1) {
            uint2 u;
            u.x = bf16_pack2(a, bq);
            u.y = bf16_pack2(c, e);
            *reinterpret_cast<uint2*>(out + (r0 + q) * D + head * 64 + d) = u;
        } else {
            f16s_store4(out + (r0 + q) * 2L * D, head * 64 + d, a, bq, c, e);
        }
    };

    if (!RES && q0 >= len) {  // whole block is padding: defined, finite output
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store_o(q0 + wave * QW + qt * 16 + fr, dt * 16 + fh * 4, 0.f, 0.f, 0.f, 0.f);
        return;
    }

    // ---- Q fragments (B operand of S^T): [qt][g][plane]
    uint4 qf[NQT][2][PLANES];
    auto load_q = [&]() {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            int q = q0 + wave * QW + qt * 16 + fr;
            q = q < qlim ? q : qlim - 1;
            const char* qp = base + (long)q * ldb;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const int c = PLANES == 1 ? (4 * g + fh) : (8 * g + 4 * pl + fh);
                    qf[qt][g][pl] = *reinterpret_cast<const uint4*>(qp + 16 * c);
                }
        }
    };

    // ---- staging geometry (LDS-DMA, lane-linear 1 KiB per wave-instruction, swizzle on the source)
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int l_row = lane / CPR, l_pos = lane % CPR;
    auto stage = [&](int kt, int st) {
        const int k0 = kt * KT16;
        const char* zero = reinterpret_cast<const char*>(g_zero16a);
#pragma unroll
        for (int j = 0; j < (NI + NW - 1) / NW; ++j) {
            const int inst = wave_u + NW * j;
            if (NI % NW != 0 && inst >= NI) break;
            const int row = inst * (64 / CPR) + l_row;
            const int key = k0 + row;
            const bool ok = key < len;
            const char* rp = base + (long)(ok ? key : 0) * ldb;
            const int kc = l_pos ^ krow_swz(row);
            const int vu = (l_pos >> 1) ^ vrow_swz(row);
            const char* ks = rp + part + 16 * kc;
            const char* vs = rp + 2 * part + 16 * ((vu << 1) | (l_pos & 1));
            glds16a(ok ? ks : zero, smem_base + st * 2 * TILE + inst * 1024);
            glds16a(ok ? vs : zero, smem_base + st * 2 * TILE + TILE + inst * 1024);
        }
    };
    auto fence = [&]() {
#if defined(ATT_ABL) && (ATT_ABL & 1)   // timing ablation only (wrong results): no workgroup barrier per key tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif defined(ATT_ABL) && (ATT_ABL & 2)  // timing ablation only: neither the DMA wait nor the barrier
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
    };

    f32x4 o[NQT][4];
    float m_run[NQT];
    // Row sums of P on the matrix pipe (the softmax is VALU-bound at head_dim 64: 250 VALU against 32 MFMA issue slots per
    // key tile in the bf16 kernel): l^T = ONES[16 x keys] P^T[keys x q] accumulates beside O^T and is rescaled with it;
    // every row of the tile holds the sum of the operands actually multiplied into O (rounded P, hi + lo for split-f16),
    // already complete over the wave — no per-element adds, no cross-lane reduction at the end.
    f32x4 lacc[NQT];
    auto init_acc = [&]() {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            m_run[qt] = -INFINITY;
            lacc[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const unsigned one2 = PLANES == 1 ? 0x3F803F80u : 0x3C003C00u;  // (1.0, 1.0) as bf16 / f16
    const uint4 ones = make_uint4(one2, one2, one2, one2);
    // Softmax exponentials: raw scores (split-f16: still carrying the operand scales 64 * 64), running maximum in
    // the same raw units, and p = 2^((s - m) * c) with c = scale * log2 e.  The subtraction comes FIRST: it is exact
    // (or rounds at the size of the difference), so the only new error is the rounding of the product, relative
    // |y| 2^-24 ln 2 in p — at most 2e-8 once weighted by p itself, the size of an f32 rounding.  (Scaling before
    // the subtraction rounds at the size of the score instead and did flip an index in a parity test.)  One
    // v_exp_f32 per element instead of libm's expf; masked keys give 2^-inf = 0.
    // split-f16: P is produced directly at its operand scale 2048 (lo halves of small probabilities stay normal)
    // by lowering the subtracted maximum by log2(2048) / c; the row sum carries the same factor and cancels it.
    const float c_exp = (PLANES == 1 ? 1.0f : 1.0f / (SWC_F16S_ACT_SCALE * SWC_F16S_ACT_SCALE)) * 1.4426950408889634f;
    const float m_off = PLANES == 1 ? 0.0f : 11.0f / c_exp;

    const int ntile = (len + KT16 - 1) / KT16;
    // one key tile; LAST: the tile may hold keys >= len (masking and the -inf guards are compiled only there)
    auto tile = [&](int kt, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const int st = RES ? kt : (kt & 1);   // RES: every tile has its own slot, staged once before the query loop
#if !(defined(ATT_ABL) && (ATT_ABL & 32))   // timing ablation only (wrong results): the next tile is not staged
        if (!RES && !LAST) stage(kt + 1, st ^ 1);
#endif
        const char* sK = smem + st * 2 * TILE;
        const char* sV = sK + TILE;
        const int k0 = kt * KT16;

        // S^T[key][q] for 4 key sub-tiles x 2 query tiles
        f32x4 s[NQT][NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int row = ks * 16 + fr;
            uint4 kf[2][PLANES];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const int c = PLANES == 1 ? (4 * g + fh) : (8 * g + 4 * pl + fh);
#if defined(ATT_ABL) && (ATT_ABL & 16)   // timing ablation only (wrong results): no LDS fragment reads
                    kf[g][pl] = qf[0][g][pl];
#else
                    kf[g][pl] = *reinterpret_cast<const uint4*>(sK + row * ROWB + ((c ^ krow_swz(row)) << 4));
#endif
                }
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if constexpr (PLANES == 2) {
                        a = mma16<PLANES>(kf[g][1], qf[qt][g][0], a);  // lo * hi
                        a = mma16<PLANES>(kf[g][0], qf[qt][g][1], a);  // hi * lo
                    }
                    a = mma16<PLANES>(kf[g][0], qf[qt][g][0], a);
                }
                s[qt][ks] = a;
            }
        }
        // mask, online softmax per query tile, P fragments
        uint4 pf[NQT][NPR][PLANES];  // [qt][pair][plane]
#if defined(ATT_ABL) && (ATT_ABL & 4)   // timing ablation only (wrong results): no softmax arithmetic, P = raw score bits
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
                    pf[qt][pr][pl] = *reinterpret_cast<uint4*>(&s[qt][(2 * pr + pl) % NKS]);
#else
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            float mt = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + ks * 16 + fh * 4 + e;
                    float v = s[qt][ks][e];
                    if (LAST) v = key < len ? v : -INFINITY;
                    s[qt][ks][e] = v;
                    mt = fmaxf(mt, v);
                }
            // the four lane groups (fh) of a query column hold partial maxima: combined with the gfx950 row swaps
            // (v_permlane16_swap / v_permlane32_swap: plain vector instructions) — __shfl_xor compiles to ds_bpermute_b32,
            // a round trip through the LDS crossbar, twice in the dependent chain max -> exp of every key tile
            // (bf16 kernel: 47.5 -> 46.2 us per launch in the pipeline; the split-f16 kernel gets another register
            // allocation with them and runs 3.6 % slower, 103.3 -> 107.0 us: it keeps the shuffles)
            if constexpr (PLANES == 1) {
                mt = xor_max<16>(mt);
                mt = xor_max<32>(mt);
            } else {
                mt = fmaxf(mt, __shfl_xor(mt, 16));
                mt = fmaxf(mt, __shfl_xor(mt, 32));
            }
            // bf16 (decode side): the reference maximum is only moved — and O, l rescaled — when some query column's
            // maximum has grown by more than 2^8 since it was last fixed (wave-uniform test).  Until then p = 2^((s - m) c)
            // may exceed 1 by up to 2^8, which bf16 P and the f32 accumulators hold without loss; the normalisation by l
            // at the end cancels the stale reference exactly.  Saves the 40 multiplies + exp of the rescale on most key
            // tiles (the kernel is VALU-bound at head_dim 64).  split-f16 (encode side, P carried at scale 2048 in f16,
            // indices must stay bit-exact) keeps the exact running maximum.
            bool rescale = true;
            if constexpr (PLANES == 1) rescale = __builtin_amdgcn_ballot_w64(mt > m_run[qt] + 8.0f / c_exp) != 0;
            float alpha = 1.0f;
            if (rescale) {
                const float m_new = fmaxf(m_run[qt], mt);
                alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * c_exp);
                m_run[qt] = m_new;
            }
            const float m_sub = m_run[qt] - m_off;
            const float mc = m_sub * c_exp;
            (void)mc;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // bf16 (decode side): one fma; split-f16 keeps the subtraction first (see above: the encoder's indices)
                    s[qt][ks][e] = PLANES == 1 ? __builtin_amdgcn_exp2f(fmaf(s[qt][ks][e], c_exp, -mc))
                                               : __builtin_amdgcn_exp2f((s[qt][ks][e] - m_sub) * c_exp);
                }
            if (rescale) {
                lacc[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[qt][dt] *= alpha;
            }
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                unsigned h[4], lo[4];
                (void)lo;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {  // pairs (2 j2, 2 j2 + 1) of the 8 keys of this k-step
                    const float p0 = s[qt][2 * pr + (j2 >> 1)][2 * (j2 & 1)], p1 = s[qt][2 * pr + (j2 >> 1)][2 * (j2 & 1) + 1];
                    if constexpr (PLANES == 1) {
                        h[j2] = pack2_bf16(p0, p1);
                    } else {
                        // p * 2048 <= 2048: no saturation clamp needed here
                        h[j2] = pack2_f16(p0, p1);
                        const f16x2 hh = *reinterpret_cast<const f16x2*>(&h[j2]);
                        lo[j2] = pack2_f16(p0 - (float)hh[0], p1 - (float)hh[1]);
                    }
                }
                pf[qt][pr][0] = make_uint4(h[0], h[1], h[2], h[3]);
                if constexpr (PLANES == 2) pf[qt][pr][1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        }
#endif
        // O^T[d][q] += V^T[d][key] P^T[key][q]
        const int tq = fr >> 2, tp = fr & 3;  // transposing read: this lane addresses row tq, columns 4 tp .. 4 tp + 3
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint4 vf[PLANES];
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const int u = PLANES == 1 ? dt : ((dt >> 1) * 4 + (dt & 1) + 2 * pl);
                    s16x4 part2[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int row = (2 * pr + t) * 16 + 4 * fh + tq;
                        const char* ap = sV + row * ROWB + ((u ^ vrow_swz(row)) << 5) + 8 * tp;
                        part2[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(uintptr_t)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)ap);
                    }
                    s16x8 full = __builtin_shufflevector(part2[0], part2[1], 0, 1, 2, 3, 4, 5, 6, 7);
                    vf[pl] = *reinterpret_cast<uint4*>(&full);
#if defined(ATT_ABL) && (ATT_ABL & 16)
                    vf[pl] = qf[0][pl & 1][0];
#endif
                }
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) {
                    f32x4 a = o[qt][dt];
                    if constexpr (PLANES == 2) {
                        a = mma16<PLANES>(vf[1], pf[qt][pr][0], a);
                        a = mma16<PLANES>(vf[0], pf[qt][pr][1], a);
                    }
                    a = mma16<PLANES>(vf[0], pf[qt][pr][0], a);
                    o[qt][dt] = a;
                }
            }
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                if constexpr (PLANES == 2) lacc[qt] = mma16<PLANES>(ones, pf[qt][pr][1], lacc[qt]);
                lacc[qt] = mma16<PLANES>(ones, pf[qt][pr][0], lacc[qt]);
            }
        if (!RES) fence();
    };
    auto finish = [&]() {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            const float l = lacc[qt][0];  // every row of the tile holds the sum for query column fr
            // split-f16: acc = sum (P * 2048) (V * 64); the output is written at the activation scale 64
            const float inv = l > 0.f ? 1.0f / l : 0.f;  // split-f16: P and its row sum both carry the factor 2048
            const int q = q0 + wave * QW + qt * 16 + fr;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = o[qt][dt] * inv;
                store_o(q, dt * 16 + fh * 4, v[0], v[1], v[2], v[3]);
            }
        }
    };
    if constexpr (RES) {
        // all K / V tiles of this (utterance, head) once, then query blocks of NW x QW queries without any further
        // staging, fence or barrier; blocks wholly beyond the length write zeros (defined, finite output)
        for (int kt = 0; kt < ntile; ++kt) stage(kt, kt);
        fence();
        for (q0 = 0; q0 < qlim; q0 += NW * QW) {
            if (q0 + wave * QW >= len || ntile == 0) {
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) store_o(q0 + wave * QW + qt * 16 + fr, dt * 16 + fh * 4, 0.f, 0.f, 0.f, 0.f);
                continue;
            }
            load_q();
            init_acc();
            for (int kt = 0; kt + 1 < ntile; ++kt) tile(kt, std::false_type{});
            tile(ntile - 1, std::true_type{});
            finish();
        }
    } else {
        load_q();
        init_acc();
        stage(0, 0);
        fence();
        for (int kt = 0; kt + 1 < ntile; ++kt) tile(kt, std::false_type{});
        tile(ntile - 1, std::true_type{});
        finish();
    }
}

#ifndef ATT_RES
#define ATT_RES 0
#endif
// the K / V-resident form (bf16, T <= 512): one workgroup of 8 waves per (utterance, head), 128 KiB of LDS
int launch16_resident(const void* qkv, void* out, const int32_t* lens, int B, int T, int H, const int32_t* row_start, hipStream_t s) {
    constexpr int LDS = 8 * 2 * 64 * 128;
    auto kern = attn16_kernel<1, 64, 2, true>;
    SWC_ENABLE_LDS(kern, LDS, "swc_attention16");
    dim3 grid(1, H, B), block(512);
    hipLaunchKernelGGL(kern, grid, block, LDS, s, (const char*)qkv, (unsigned short*)out, lens, T, H, row_start);
    return SWC_OK;
}

template <int PLANES, int KT16, int NQT>
int launch16(const void* qkv, void* out, const int32_t* lens, int B, int T, int H, const int32_t* row_start, hipStream_t s) {
    constexpr int LDS = 2 * 2 * KT16 * 128 * PLANES;
    auto kern = attn16_kernel<PLANES, KT16, NQT>;
    constexpr int QB16 = 64 * NQT;
    if (LDS > 48 * 1024) SWC_ENABLE_LDS(kern, LDS, "swc_attention16");
    dim3 grid((T + QB16 - 1) / QB16, H, B), block(256);
    hipLaunchKernelGGL(kern, grid, block, LDS, s, (const char*)qkv, (unsigned short*)out, lens, T, H, row_start);
    return SWC_OK;
}

}  // namespace

// dtype: SWC_BF16 (bf16 in, bf16 out) or SWC_F16S (split-f16 in at scale 64, split-f16 out at scale 64)
extern "C" int swc_attention16(const void* qkv, void* out, const int32_t* lens, int32_t B, int32_t T, int32_t H,
                               int32_t dtype, const int32_t* row_start, void* stream) {
    SWC_CHECK_ARG(qkv && out && lens, "swc_attention16: null pointer");
    SWC_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && B <= 65535 && H <= 65535, "swc_attention16: bad B/T/H");
    SWC_CHECK_ARG(dtype == SWC_BF16 || dtype == SWC_F16S, "swc_attention16: dtype must be BF16 or F16S");
    SWC_CHECK_ARG(aligned16(qkv) && aligned16(out), "swc_attention16: unaligned");
    if (B == 0 || T == 0) return SWC_OK;
    // Measured alternatives for the bf16 kernel (tools/bench_attention.py, T = 500): 128-key tiles (half the fences per key,
    // occupancy 3 -> 2) 56.2 us against 54.9; 16 queries per wave (91 registers, occupancy 5, but every K / V fragment
    // feeds half the MFMAs) 70.5 us against 56.0.  PMC (tools/pmc_attention.sh): the SIMDs spend 58 % of their cycles
    // issuing (VALU 46 %), matrix pipe 22 % busy, no LDS bank conflict: the kernel is bound by instructions per score.
    int rc = dtype == SWC_BF16 ? ((ATT_RES && T <= 512) ? launch16_resident(qkv, out, lens, B, T, H, row_start, (hipStream_t)stream)
                                                        : launch16<1, 64, 2>(qkv, out, lens, B, T, H, row_start, (hipStream_t)stream))
                               : launch16<2, 64, 2>(qkv, out, lens, B, T, H, row_start, (hipStream_t)stream);
    if (rc != SWC_OK) return rc;
    SWC_CHECK_LAUNCH("swc_attention16");
    return SWC_OK;
}
