// swc_gemm: C = epi(A (*) W^T) on gfx950 MFMA — plain GEMM or implicit-GEMM Conv1d over
// frame-major activations.
//
// One template, two geometries (WAVES_M x WAVES_N waves, each wave 16*MT x 64 outputs):
//   128 x 128 tile, 4 waves (2x2, MT=4), 64 KiB LDS, 2 workgroups / CU   — f32 and bf16, any shape
//   256 x 256 tile, 8 waves (2x4, MT=8), 128 KiB LDS, 1 workgroup / CU   — bf16, large M and N
// K is walked in 128-byte slices (32 f32 / 64 bf16) through a double-buffered LDS image filled by
// LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write).
//
// f32 : v_mfma_f32_16x16x4_f32   (bit-for-bit an f32 fma chain: the parity path)
// bf16: v_mfma_f32_16x16x32_bf16 (f32 accumulate)
//
// LDS image: row r (128 bytes) holds eight 16-byte chunks; logical chunk c sits at position
// c ^ swz(r).  One LDS-DMA wave-instruction writes 8 rows lane-linearly, so the swizzle is applied to
// the per-lane SOURCE address.  Fragment reads (ds_read_b128): lane (fr = l & 15, fh = l >> 4) reads
// chunk fh + 4g of its row; for f32 the chunk's 4 floats feed 4 successive MFMA k-steps (a permuted but
// complete walk of the slice), for bf16 the chunk is the 8-element operand of one 16x16x32 step.
//
// The product is computed TRANSPOSED (weight fragment = MFMA row operand) with the weight rows of a
// fragment taken as {16a + 4j + b}: lane (fr, fh) then owns, for activation row 16i + fr, the 16
// CONTIGUOUS output columns 16fh + 4j + e — 64-byte vector stores, full lines per row across fh.
#include "swc_common.h"

// Geometry overrides for A/B measurements exist only in tuning builds (-DSWC_TUNING, tools/build_variant.sh): the
// shipped library reads no environment variable and keeps no mutable process-wide state.
static inline int tuning_env(const char* name, int dflt = 0) {
#ifdef SWC_TUNING
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

namespace {


__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};
#ifdef SWC_GEMM_STAMP
__device__ unsigned long long g_stamps[512 * 8 * 8];  // [workgroup][wave][compute, vmcnt wait, barrier wait, slices, epilogue, tiles, tail, -]
#endif

// 16 bytes per lane, global -> LDS (global_load_lds_dwordx4).  `lds_addr` is the wave-uniform LDS byte
// address; lane l lands at lds_addr + 16 l.  Written as inline asm on purpose: hipcc drains a
// compiler-visible LDS-DMA (s_waitcnt vmcnt(0)) in front of the next ds_read, which would serialise the
// prefetch of slice t+1 with the MFMAs of slice t.  Hidden in asm, the DMA is ordered by our own
// `s_waitcnt vmcnt(N)` + barrier (cdna_hip_programming.md 5.7).  M0 is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);  // wave-uniform by construction; pin it to an SGPR
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}
// same, address = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: one VGPR per lane
// instead of a 64-bit pointer per staged row
__device__ __forceinline__ void glds16_s(const char* base, unsigned off, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(base), "s"(lds_addr)
        : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// conflict-free for the four 16-lane groups of ds_read_b128, both for 16 consecutive rows (activation
// fragments) and for the permuted weight rows {16a + 4j + b} (checked exhaustively)
__device__ __forceinline__ int swz(int row) { return ((row >> 1) ^ ((row >> 4) << 1)) & 7; }
// 64-byte rows: conflict-free for both patterns as well (checked exhaustively)
__device__ __forceinline__ int swz64(int row) { return ((-(row >> 2)) ^ (-(row >> 4))) & 3; }
template <int RB>
__device__ __forceinline__ int swz_rb(int row) { return RB == 128 ? swz(row) : swz64(row); }
template <int RB>
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * RB + ((chunk ^ swz_rb<RB>(row)) << 4); }


struct GemmP {
    const char* A;
    const char* W;
    void* C;
    const float* bias;
    const float* gamma;
    const float* residual;
    long lda, ldw, ldc, ldr;  // elements
    int M, N, K;
    int taps, dil, stride, pad, t_in, t_out;
    int act;
    float alpha, out_scale;
    int band;        // tile-order band height (row panels), see the kernel
    unsigned* sat;   // saturation counter pair of the calling thread (swc_set_saturation_counter) or nullptr
    int kc_per_tap;  // ceil(K / BK)
    int n_tiles_n, n_tiles_m;
    int nt_mode;     // epilogue stores non-temporal: 0 never, 1 on the LAST tile of a workgroup, 2 always (swc_gemm chooses)
    int stagger;     // start delay unit in shader cycles: workgroup w starts (w / 8 % 8) * stagger cycles late (0: none)
};

// Epilogue shared by all geometries.  Transposed product: lane (fr, fh) holds, for activation row
// 16i + fr of its wave's slab, the 16 contiguous output columns 16fh + 4j + e.
constexpr int EPI_TP = 68;                       // row pitch (floats) of a wave's transpose block: conflict-free b128 writes
constexpr int EPI_WAVE_BYTES = 16 * EPI_TP * 4;  // 16 rows x 64 columns per wave


// Epilogue stores.  Tuning builds can change how they are issued (tools/build_variant.sh):
//   -DSWC_ABL_NOSTORE   timing ablation only, WRONG RESULTS: the values are computed and kept live, nothing is stored —
//                       what a launch would cost if its epilogue stores were free
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// `nt` (wave-uniform): non-temporal store — the line is not kept dirty in the XCD's L2 (see GemmP::nt_mode)
__device__ __forceinline__ void epi_st128(void* p, unsigned a, unsigned b, unsigned c, unsigned d, bool nt = false) {
#if defined(SWC_ABL_NOSTORE)
    asm volatile("" ::"v"(p), "v"(a), "v"(b), "v"(c), "v"(d));
#elif defined(SWC_EPI_PLAIN)
    (void)nt;
    *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#else
    if (nt) __builtin_nontemporal_store((u32x4_t){a, b, c, d}, reinterpret_cast<u32x4_t*>(p));
    else *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#endif
}
__device__ __forceinline__ void epi_st128f(void* p, float a, float b, float c, float d, bool nt = false) {
    epi_st128(p, __float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d), nt);
}
__device__ __forceinline__ void epi_st64(void* p, unsigned a, unsigned b, bool nt = false) {
#if defined(SWC_ABL_NOSTORE)
    asm volatile("" ::"v"(p), "v"(a), "v"(b));
#elif defined(SWC_EPI_PLAIN)
    (void)nt;
    *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
#else
    if (nt) __builtin_nontemporal_store((u32x2_t){a, b}, reinterpret_cast<u32x2_t*>(p));
    else *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
#endif
}
__device__ __forceinline__ void epi_st32(void* p, unsigned a, bool nt = false) {
#if defined(SWC_ABL_NOSTORE)
    asm volatile("" ::"v"(p), "v"(a));
#elif defined(SWC_EPI_PLAIN)
    (void)nt;
    *reinterpret_cast<unsigned*>(p) = a;
#else
    if (nt) __builtin_nontemporal_store(a, reinterpret_cast<unsigned*>(p));
    else *reinterpret_cast<unsigned*>(p) = a;
#endif
}

template <typename OutT>
__device__ __forceinline__ float epi_act(float x) {
    return (__is_same(OutT, bf16_t) || __is_same(OutT, fp8_t)) ? gelu_fast(x) : (__is_same(OutT, f16s_t) ? gelu_as(x) : gelu_erf(x));
}

// Coalesced epilogue for a wave whose 64-column slab lies inside N: every 16 x 64 accumulator block goes through the
// wave's own LDS scratch (no workgroup barrier: one wave, LDS operations of a wave complete in order) and comes back
// with lane l on row (l >> 4) + 4k, columns 4 (l & 15) .. + 3: one wave-instruction then covers 4 rows x 256 contiguous
// bytes (f32), the residual is read and the output written in whole 64-byte sectors.  In the accumulator layout a lane
// owns 16 contiguous columns of ONE row, so one instruction touched 64 separate 16-byte pieces: the f32 epilogue of a
// 192 x 256 tile took 24k cycles on an idle chip (40k with every CU in its epilogue: 60 % of the out-proj GEMM,
// profiles/r02_gemm_stamps.txt).  Same arithmetic per element, in the same order, as the direct path below.
template <typename OutT, int MT, bool GELU>
__device__ __forceinline__ void gemm_epilogue_slab(const GemmP& p, f32x4 (&acc)[MT][4], int row0, int colw, int fr, int fh,
                                                   float* tbuf, float& amax, bool nt) {
    int lane = fr + 16 * fh;
    // opaque to the optimiser: everything below depends on it and is therefore computed HERE, once per tile; hoisted out
    // of the persistent tile loop these per-lane addresses stayed live through the K loop, which has no register to spare
    // (they were spilled and re-loaded in every slice)
    asm volatile("" : "+v"(lane));
    const int tr = lane >> 4, tc = (lane & 15) * 4;
    const int col = colw + tc;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + col);
    if (p.gamma) g4 = *reinterpret_cast<const float4*>(p.gamma + col);
    float* wr_p = tbuf + (lane & 15) * EPI_TP + (lane >> 4) * 16;
    const float* rd_p = tbuf + tr * EPI_TP + tc;
    // The residual rows of block i + RD are requested before block i is processed (a rolling window of RD + 1 blocks in
    // registers).  C may alias the residual (the residual stream is updated in place), so hipcc keeps every load behind the
    // stores that precede it in program order: loaded block by block, each of the MT blocks waited a full memory latency with
    // every CU doing the same (26 - 33 k cycles per 192 x 256 tile, in-kernel stamps).  All MT blocks at once (16 MT
    // registers) made the K loop spill; RD blocks ahead cost 16 RD registers that are free in the epilogue.
#ifndef SWC_EPI_RD
#define SWC_EPI_RD 2
#endif
    constexpr int RD = SWC_EPI_RD < MT ? SWC_EPI_RD : MT - 1;
    float4 r4[MT][4];
    auto load_res = [&](int i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = row0 + 16 * i + tr + 4 * k;
            r4[i][k] = row < p.M ? *reinterpret_cast<const float4*>(p.residual + (long)row * p.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (p.residual) {
#pragma unroll
        for (int i = 0; i < RD; ++i) load_res(i);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int rbase = row0 + 16 * i + tr;
        if (p.residual && i + RD < MT) load_res(i + RD);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(wr_p + 4 * j) = acc[i][j];
        f32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const f32x4*>(rd_p + 4 * k * EPI_TP);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = rbase + 4 * k;
            float v0 = t[k][0] * p.alpha + b4.x, v1 = t[k][1] * p.alpha + b4.y, v2 = t[k][2] * p.alpha + b4.z, v3 = t[k][3] * p.alpha + b4.w;
            if constexpr (GELU) { v0 = epi_act<OutT>(v0); v1 = epi_act<OutT>(v1); v2 = epi_act<OutT>(v2); v3 = epi_act<OutT>(v3); }
            v0 *= g4.x; v1 *= g4.y; v2 *= g4.z; v3 *= g4.w;
            if (p.residual) { v0 += r4[i][k].x; v1 += r4[i][k].y; v2 += r4[i][k].z; v3 += r4[i][k].w; }
            if (row >= p.M) continue;
            if constexpr (__is_same(OutT, f16s_t)) {
                const float os = p.out_scale;
                f16s_store4(reinterpret_cast<unsigned short*>(p.C) + (long)row * p.ldc * 2, col, v0 * os, v1 * os, v2 * os, v3 * os, amax);
            } else if constexpr (sizeof(OutT) == 4) {
                epi_st128f(reinterpret_cast<float*>(p.C) + (long)row * p.ldc + col, v0, v1, v2, v3, nt);
            } else if constexpr (sizeof(OutT) == 1) {
                const float os = p.out_scale;
                epi_st32(reinterpret_cast<unsigned char*>(p.C) + (long)row * p.ldc + col, fp8_pack4(v0 * os, v1 * os, v2 * os, v3 * os, amax), nt);
            } else {
                epi_st64(reinterpret_cast<bf16_t*>(p.C) + (long)row * p.ldc + col, bf16_pack2(v0, v1), bf16_pack2(v2, v3), nt);
            }
        }
    }
}

// Direct epilogue (slabs cut by N, unaligned pointers, split-f16 outputs): lane (fr, fh) holds, for row 16i + fr, columns
// 16fh + 4j + e of its wave's 64-column slab.
template <typename OutT, int MT, int GELU>  // GELU: 0 none, 1 always, 2 by p.act (one body)
__device__ __forceinline__ void gemm_epilogue_direct(const GemmP& p, f32x4 (&acc)[MT][4], int bm, int bn, int wr, int wc, int fr,
                                                     int fh, bool nt) {
    OutT* C = reinterpret_cast<OutT*>(p.C);
    const int col0 = bn + wc * 64 + 16 * fh;
    const bool vec_ok = (col0 + 16 <= p.N) && ((p.ldc & (sizeof(OutT) == 1 ? 15 : 3)) == 0) &&
                        (!p.residual || (p.ldr & 3) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0);
    // gamma (ConvNeXt layer scale) exists for f32 outputs only (swc_gemm checks): the 16-bit-output bodies do not carry a
    // multiply by 1.0 per output (bf16 fc1 / qkv -3 %, profiles/r03_gemm_split_ab.txt).  History: under hipcc's default
    // scheduler compiling it out flipped the split-f16 fc1 kernel into a K-loop schedule that ran 11 % slower (226 against
    // 204 us, profiles/r03_gemm_epilogue_ab.txt); with the max-ilp scheduler and the explicit fence split the K loop no longer
    // depends on the epilogue (tools/check_isa.py watches its shape)
#ifdef SWC_KEEP_GAMMA16
    constexpr bool HAS_GAMMA = true;
#else
    constexpr bool HAS_GAMMA = sizeof(OutT) == 4;
#endif
    float bv[16], gv[HAS_GAMMA ? 16 : 1];
    float amax = 0.f;
    (void)amax; (void)gv;
    // a lane's 16 columns are 64 contiguous bytes of bias / gamma: four 16-byte loads when they lie inside N (one load and
    // one branch per column otherwise: 32 dependent-looking scalar loads per tile on the hot GEMMs)
    if (col0 + 16 <= p.N && ((reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.gamma)) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + col0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
            if constexpr (HAS_GAMMA) {
                const float4 g4 = p.gamma ? *reinterpret_cast<const float4*>(p.gamma + col0 + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
                gv[4 * q] = g4.x; gv[4 * q + 1] = g4.y; gv[4 * q + 2] = g4.z; gv[4 * q + 3] = g4.w;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int col = col0 + c;
            bv[c] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
            if constexpr (HAS_GAMMA) gv[c] = (p.gamma && col < p.N) ? p.gamma[col] : 1.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = bm + wr * (MT * 16) + i * 16 + fr;
        if (row >= p.M) continue;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[i][j][e] * p.alpha + bv[4 * j + e];
        if constexpr (GELU == 1) {
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = epi_act<OutT>(v[c]);
        } else if constexpr (GELU == 2) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (p.act == SWC_ACT_GELU) v[c] = epi_act<OutT>(v[c]);
        }
        if constexpr (HAS_GAMMA) {
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] *= gv[c];
        }
        if constexpr (sizeof(OutT) == sizeof(f16s_t) && !__is_same(OutT, bf16_t)) {
            // split-f16 output: the lane's 16 columns sit inside one 32-block (col0 % 16 == 0, N % 32 == 0)
            if (col0 + 16 <= p.N) {
                if (p.residual) {
                    const float4* rp = reinterpret_cast<const float4*>(p.residual + (long)row * p.ldr + col0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 r4 = rp[j];
                        v[4 * j] += r4.x; v[4 * j + 1] += r4.y; v[4 * j + 2] += r4.z; v[4 * j + 3] += r4.w;
                    }
                }
                unsigned short* rowp = reinterpret_cast<unsigned short*>(p.C) + (long)row * p.ldc * 2;
                unsigned h2[8], l2[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) f16s_split2(v[2 * c] * p.out_scale, v[2 * c + 1] * p.out_scale, h2[c], l2[c], amax);
                unsigned short* hp = rowp + f16s_col(col0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    epi_st128(reinterpret_cast<uint4*>(hp) + q, h2[4 * q], h2[4 * q + 1], h2[4 * q + 2], h2[4 * q + 3], nt);
                    epi_st128(reinterpret_cast<uint4*>(hp + 32) + q, l2[4 * q], l2[4 * q + 1], l2[4 * q + 2], l2[4 * q + 3], nt);
                }
            }
        } else if (vec_ok) {
            if (p.residual) {
                const float4* rp = reinterpret_cast<const float4*>(p.residual + (long)row * p.ldr + col0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 r4 = rp[j];
                    v[4 * j] += r4.x; v[4 * j + 1] += r4.y; v[4 * j + 2] += r4.z; v[4 * j + 3] += r4.w;
                }
            }
            OutT* cp = C + (long)row * p.ldc + col0;
            if constexpr (sizeof(OutT) == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    epi_st128f(reinterpret_cast<float4*>(cp) + j, v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3], nt);
            } else if constexpr (sizeof(OutT) == 1) {
                const float os = p.out_scale;
                const unsigned u0 = fp8_pack4(v[0] * os, v[1] * os, v[2] * os, v[3] * os, amax);
                const unsigned u1 = fp8_pack4(v[4] * os, v[5] * os, v[6] * os, v[7] * os, amax);
                const unsigned u2 = fp8_pack4(v[8] * os, v[9] * os, v[10] * os, v[11] * os, amax);
                const unsigned u3 = fp8_pack4(v[12] * os, v[13] * os, v[14] * os, v[15] * os, amax);
                epi_st128(cp, u0, u1, u2, u3, nt);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    epi_st128(reinterpret_cast<uint4*>(cp) + j, bf16_pack2(v[8 * j + 0], v[8 * j + 1]), bf16_pack2(v[8 * j + 2], v[8 * j + 3]),
                              bf16_pack2(v[8 * j + 4], v[8 * j + 5]), bf16_pack2(v[8 * j + 6], v[8 * j + 7]), nt);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int col = col0 + c;
                if (col >= p.N) continue;
                float x = v[c];
                if (p.residual) x += p.residual[(long)row * p.ldr + col];
                if constexpr (__is_same(OutT, fp8_t)) { x *= p.out_scale; amax = fmaxf(amax, fabsf(x)); }
                if constexpr (!__is_same(OutT, f16s_t)) store_out<OutT>(C + (long)row * p.ldc + col, x);
            }
        }
    }
    if constexpr (__is_same(OutT, f16s_t)) sat_commit(p.sat, 0, amax, SWC_F16S_LIMIT);
    if constexpr (__is_same(OutT, fp8_t)) sat_commit(p.sat, 1, amax, SWC_FP8_LIMIT);
}

// The activation is a template parameter behind a wave-uniform branch: as a run-time select inside one body hipcc evaluates
// the GELU of every element of every GEMM and selects afterwards — the GEMMs without an activation (qkv, out-proj, fc2,
// pwconv2, every conv) paid 10 (bf16) to 40 (erff) VALU instructions per output for nothing (tools/gemm_stamps.py).
// ACT: 0 no activation, 1 GELU (both compiled in: one body per kernel — the 16-bit-output plain GEMMs, whose K loops have no
// register to spare for a second epilogue body), 2 by p.act at run time (wave-uniform branch over two bodies).
template <typename OutT, int MT, bool SLAB, int ACT>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[MT][4], int bm, int bn, int wr, int wc, int fr,
                                              int fh, float* tbuf, bool nt) {
    const bool gelu = ACT == 1 || (ACT == 2 && p.act == SWC_ACT_GELU);
    // f32 outputs (residual stream: out-proj, fc2, pwconv2, heads) take the transposed path: +9 ... +15 % on those GEMMs.
    // 16-bit outputs keep the direct one: a lane's 16 columns are 32 contiguous bytes there, the transposed path measured
    // no faster without and 6 - 8 % slower with the GELU, and split-f16 conversion leaves no register for it
    if constexpr (SLAB && sizeof(OutT) == 4 && !__is_same(OutT, f16s_t)) {
        const int colw = bn + wc * 64;
        const bool al16 = ((reinterpret_cast<uintptr_t>(p.C) | reinterpret_cast<uintptr_t>(p.residual) |
                            reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.gamma)) & 15) == 0;
        const bool ok = tbuf && colw + 64 <= p.N && al16 && (p.ldc & (sizeof(OutT) == 1 ? 15 : 3)) == 0 &&
                        (!p.residual || (p.ldr & 3) == 0);
        if (__builtin_amdgcn_readfirstlane((int)ok)) {  // wave-uniform by construction
            float amax = 0.f;
            if (gelu) gemm_epilogue_slab<OutT, MT, true>(p, acc, bm + wr * (MT * 16), colw, fr, fh, tbuf, amax, nt);
            else gemm_epilogue_slab<OutT, MT, false>(p, acc, bm + wr * (MT * 16), colw, fr, fh, tbuf, amax, nt);
            if constexpr (__is_same(OutT, fp8_t)) sat_commit(p.sat, 1, amax, SWC_FP8_LIMIT);
            return;
        }
    }
    if constexpr (ACT != 2) {
        gemm_epilogue_direct<OutT, MT, ACT>(p, acc, bm, bn, wr, wc, fr, fh, nt);
    } else if constexpr (__is_same(OutT, f16s_t) && MT == 8) {
        // the 256-row split-f16 conv kernel has no register for two bodies (the K loop spilled): one body, GELU by p.act
        gemm_epilogue_direct<OutT, MT, 2>(p, acc, bm, bn, wr, wc, fr, fh, nt);
    } else {
        if (gelu) gemm_epilogue_direct<OutT, MT, 1>(p, acc, bm, bn, wr, wc, fr, fh, nt);
        else gemm_epilogue_direct<OutT, MT, 0>(p, acc, bm, bn, wr, wc, fr, fh, nt);
    }
}

// PLAIN: one tap, K a multiple of the slice, no stride / padding / row remap — the hot GEMMs.  A separate
// instantiation so that the implicit-conv bookkeeping (~32 VGPRs) does not sit in the hot loop's register budget.
template <int MODE, typename OutT, int MT, int WAVES_M, int WAVES_N, bool PLAIN, int RB = 128, int ACT = 2>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_kernel(GemmP p) {
    constexpr int ROW_BYTES = RB;  // bytes of K per LDS row per slice: 128, or 64 (bf16 / f32 only: smaller stages, two workgroups per CU)
    constexpr int CPR = RB / 16;   // 16-byte chunks per row
    constexpr int NG = RB / 64;    // 64-byte k-groups per slice
    static_assert(RB == 128 || (RB == 64 && MODE != SWC_F16S), "split-f16 needs hi|lo in one 128-byte row");
    static_assert(WAVES_M * WAVES_N * EPI_WAVE_BYTES <= (WAVES_M * MT * 16 + WAVES_N * 64) * RB, "epilogue scratch must fit one stage");
    constexpr bool BF16 = MODE == SWC_BF16;
    constexpr bool F16S = MODE == SWC_F16S;
    constexpr bool FP8 = MODE == SWC_FP8;
    constexpr int ES = BF16 ? 2 : (FP8 ? 1 : 4);  // bytes per LOGICAL element (split-f16 is 2 halves = 4 bytes)
    constexpr int EPC = 16 / ES;        // logical elements per 16-byte chunk (K-tail predicate only)
    constexpr int BK = ROW_BYTES / ES;  // logical elements of K per slice
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BM = WAVES_M * MT * 16;
    constexpr int BN = WAVES_N * 64;
    constexpr int RPS = NT / CPR;  // rows covered by one staging sweep of the workgroup
    constexpr int NA = BM / RPS;  // A chunks per thread per slice
    constexpr int NB = BN / RPS;  // W chunks per thread per slice
    constexpr int A_BYTES = BM * ROW_BYTES;
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [stage][A rows | W rows][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    // XCD-aware, persistent tile walk: workgroups b and b+8 share an XCD (round-robin dispatch), so every XCD owns
    // a contiguous chunk of the tile sequence (neighbours share operand panels in its L2) and its workgroups stride
    // through that chunk.  With gridDim == number of tiles every workgroup gets exactly one tile.
    const int ntiles = p.n_tiles_m * p.n_tiles_n;
    const int xcd = blockIdx.x & 7;
    const int gw = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups on this XCD
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk_start = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunk_size = cq + (xcd < cr ? 1 : 0);
    int tl = blockIdx.x >> 3;  // local tile index inside the chunk
    if (tl >= chunk_size) return;
    int bm, bn;
    // tiles are walked in bands of p.band row panels, column-major inside a band: with band = 4 the ~32 tiles an XCD
    // runs at the same time form a 4 x 8 block (12 distinct operand panels per K step instead of 18 for a 2 x 16
    // strip: +9 % at 32768 x 4096 x 4096); narrow outputs (few column tiles) keep band = 1 (row-major)
    auto tile_origin = [&](int local) {
        const int bid = chunk_start + local;
        const int BAND = p.band;
        const int band = bid / (BAND * p.n_tiles_n);
        const int rem = bid - band * (BAND * p.n_tiles_n);
        const int rows_in_band = (p.n_tiles_m - band * BAND) < BAND ? (p.n_tiles_m - band * BAND) : BAND;
        const int tn = rem / rows_in_band;
        const int tm = band * BAND + (rem - tn * rows_in_band);
        bm = tm * BM;
        bn = tn * BN;
    };

    // ---- per-thread staging geometry (per tile: setup_tile)
    const int ld_pos = tid % CPR;  // chunk POSITION this lane lands on
    const int ld_row = tid / CPR;  // row inside one staging sweep
    int a_b[NA], a_t[NA], a_chunk[NA];
    bool a_rowok[NA];
    long w_rowoff[NB];
    int w_chunk[NB];
    bool w_rowok[NB];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Lanes whose tap / K-chunk is out of range fetch a 16-byte device zero page (LDS-DMA cannot
    // zero-fill).  Rows beyond M / N are clamped instead (their results are never stored).
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    // PLAIN staging: address = wave-uniform tile base (SGPRs) + one 32-bit offset per lane.  Row r of a sweep sits at
    // r * ld_bytes + (chunk << 4); the swizzled chunk repeats every 64 rows, so NV (1 or 2) per-lane constants cover
    // all sweeps, and rows beyond M / N are clamped to the last valid row (their results are never stored).
    constexpr int NV = RPS >= 64 ? 1 : 64 / RPS;
    unsigned chunkoff[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) chunkoff[v] = (unsigned)(ld_pos ^ swz_rb<RB>(ld_row + RPS * v)) << 4;
    const unsigned lda_bytes = (unsigned)(p.lda * ES), ldw_bytes = (unsigned)(p.ldw * ES);  // < 2^24 (host check)
    const char* a_base = p.A;
    const char* w_base = p.W;
    int a_left = 1, w_left = 1;
    auto setup_tile = [&](int local) {
        tile_origin(local);
        if constexpr (PLAIN) {
            a_base = p.A + (long)bm * lda_bytes;
            w_base = p.W + (long)bn * ldw_bytes;
            a_left = p.M - bm;
            w_left = p.N - bn;
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = ld_row + RPS * i;
            const int r = bm + row;
            const int chunk = ld_pos ^ swz_rb<RB>(row);
            if constexpr (!PLAIN) {
                a_rowok[i] = r < p.M;
                const int rr = a_rowok[i] ? r : 0;
                a_b[i] = rr / p.t_out;
                a_t[i] = rr - a_b[i] * p.t_out;
                a_chunk[i] = chunk;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = ld_row + RPS * i;
            const int n = bn + row;
            const int chunk = ld_pos ^ swz_rb<RB>(row);
            if constexpr (!PLAIN) {
                w_rowok[i] = n < p.N;
                w_rowoff[i] = (long)(w_rowok[i] ? n : 0) * p.ldw;
                w_chunk[i] = chunk;
            }
        }
    };
    setup_tile(tl);
    // De-phasing (multi-tile launches): every workgroup runs the same K loop + epilogue sequence, so all 256 CUs enter
    // their HBM-bound epilogue together and the matrix pipes of the whole chip idle while the fabric drains 64 MB;
    // started a fraction of an epilogue apart, the epilogues of one phase group fall into the K loops of the others.
    if (p.stagger > 0) {
        const int phase = (blockIdx.x >> 3) & 7;
        if (phase) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            const unsigned long long wait = (unsigned long long)phase * (unsigned)p.stagger;
            while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        }
    }
    auto stage_slice = [&](int kt, int stage) {
        const unsigned sa = smem_base + stage * STAGE_BYTES + wave_u * 1024;  // one wave-instruction = 1 KiB of rows
        const unsigned sb = sa + A_BYTES;
        // (round 2 kept a never-taken uniform branch here: under hipcc's default scheduler it made the DMA block its own
        // scheduling region and was worth 6.5 % of the step.  With -amdgpu-sched-strategy=max-ilp (build.py) the K loop has the
        // same shape with and without it — tools/check_isa.py, profiles/r03_sched_strategy_ab.txt — and it is gone.)
        if constexpr (PLAIN) {
            const long koff = (long)kt * ROW_BYTES;
            const char* ab = a_base + koff;
            const char* wb = w_base + koff;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = min(ld_row + RPS * i, a_left - 1);
                glds16_s(ab, __umul24(r, lda_bytes) + chunkoff[i % NV], sa + RPS * i * ROW_BYTES);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = min(ld_row + RPS * i, w_left - 1);
                glds16_s(wb, __umul24(r, ldw_bytes) + chunkoff[i % NV], sb + RPS * i * ROW_BYTES);
            }
            return;
        }
        const int tap = kt / p.kc_per_tap;
        const int kc = kt - tap * p.kc_per_tap;
        const int shift = tap * p.dil - p.pad;
        const char* zero = reinterpret_cast<const char*>(g_zero16);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k0 = kc * BK + a_chunk[i] * EPC;
            const int ts = a_t[i] * p.stride + shift;
            const bool ok = a_rowok[i] && k0 < p.K && ts >= 0 && ts < p.t_in;
            const char* src = p.A + (((long)a_b[i] * p.t_in + (ok ? ts : 0)) * p.lda + (ok ? k0 : 0)) * ES;
            glds16(ok ? src : zero, sa + RPS * i * ROW_BYTES);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k0 = kc * BK + w_chunk[i] * EPC;
            const bool ok = w_rowok[i] && k0 < p.K;
            const char* src = p.W + (w_rowoff[i] + (long)tap * p.K + (ok ? k0 : 0)) * ES;
            glds16(ok ? src : zero, sb + RPS * i * ROW_BYTES);
        }
    };
    // the LDS-DMA is invisible to the compiler: order it ourselves
#ifdef SWC_GEMM_STAMP
    // diagnostic build only (tools/gemm_stamps.py): where the waves of the K loop wait.  Stamps go to g_stamps, which no
    // other code reads.
    unsigned long long st_last = 0, st_cmp = 0, st_vm = 0, st_bar = 0, st_n = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    auto dma_fence = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = stamp();
        __syncthreads();
        const unsigned long long t3 = stamp();
        if (st_last) { st_cmp += t1 - st_last; st_vm += t2 - t1; st_bar += t3 - t2; st_n += 1; }
        st_last = t3;
    };
#else
    auto dma_fence = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
#endif

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nkt = p.taps * p.kc_per_tap;
    const int fr = lane & 15, fh = lane >> 4;
    const int a_row0 = wr * (MT * 16) + fr;
    const int b_row0 = wc * 64 + 16 * (fr >> 2) + (fr & 3);
    stage_slice(0, 0);
    dma_fence();
#ifdef SWC_GEMM_PRIO47   // tuning experiment: static priority for the second wave of every SIMD (MI355X_MICROARCH.md item 4)
    if (wave_u >= (WAVES_M * WAVES_N) / 2) __builtin_amdgcn_s_setprio(1);
#endif
    for (;;) {
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage_slice(kt + 1, cur ^ 1);
#ifdef SWC_GEMM_IGLP     // tuning experiment: LLVM's MFMA / DS interleaving pipelines for this scheduling region
        __builtin_amdgcn_iglp_opt(SWC_GEMM_IGLP);
#endif
        const char* sa = smem + cur * STAGE_BYTES;
        const char* sb = sa + A_BYTES;
        // 8-wave bf16 / split-f16 geometries: the slice's fragments go to registers first, and the MFMAs of the LAST `SPLIT` row
        // blocks are issued BEHIND the slice fence: they need registers only, so they fill the matrix pipe while the waves
        // come out of the barrier, issue the next slice's LDS-DMA and wait for its first fragment reads.  hipcc moves MFMAs
        // across a barrier at will (they touch no memory) and its choice changed with unrelated edits (25 ... 59 of 96 behind,
        // 204 ... 226 us per launch of the same GEMM); here the split is explicit and pinned by scheduling barriers.  Measured
        // per kernel, same box, us per launch with 0 / 1 / 2 / 3 row blocks behind the fence (profiles/r03_gemm_split_ab.txt):
        //   split-f16 256-row (fc1)  223.0 / 209.0 / 205.1 / 202.1     192-row f32-out 134.2 / 124.0 / 127.6 / 129.5, qkv 160.4 / 151.3 / 152.4 / 152.8
        //   bf16      256-row (fc1)   97.6 /  95.8 /  95.5 /  95.3     192-row f32-out  64.3 /  64.3 /  66.4 /  67.8, qkv  74.0 /  73.5 /  73.3 /  74.4
        // Same products in the same order per accumulator: results bit-identical to the plain loop below.
        // The bf16 kernels keep the compiler's own placement (3 - 5 MFMAs behind the fence under max-ilp): none of the explicit
        // splits beat it there.
#ifdef SWC_GEMM_SPLIT
        constexpr bool EXPLICIT_SPLIT = (BF16 || F16S) && WAVES_M * WAVES_N == 8 && MT >= 4;
        constexpr int SPLIT = SWC_GEMM_SPLIT;
#else
        // The 4-wave geometries (two workgroups per CU, out of phase by themselves; the GEMMs of small batches) take the
        // same body with NOTHING behind the fence: B = 8 x 10 s `mixed` 9.46 -> 9.22 ms per step, `bf16` 7.38 -> 7.34; one
        // row block behind it is slower (9.50 / 7.47).
        constexpr bool EXPLICIT_SPLIT = (F16S && WAVES_M * WAVES_N == 8 && MT >= 4) || ((BF16 || F16S) && WAVES_M * WAVES_N == 4);
        constexpr int SPLIT = WAVES_M * WAVES_N == 4 ? 0 : (MT == 8 ? 3 : 1);
#endif
        if constexpr (EXPLICIT_SPLIT) {
            uint4 f0a[MT], f0b[4], f1a[MT], f1b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f0b[j] = *reinterpret_cast<const uint4*>(sb + lds_off<RB>(b_row0 + 4 * j, fh));
#pragma unroll
            for (int i = 0; i < MT; ++i) f0a[i] = *reinterpret_cast<const uint4*>(sa + lds_off<RB>(a_row0 + 16 * i, fh));
#pragma unroll
            for (int j = 0; j < 4; ++j) f1b[j] = *reinterpret_cast<const uint4*>(sb + lds_off<RB>(b_row0 + 4 * j, fh + 4));
#pragma unroll
            for (int i = 0; i < MT; ++i) f1a[i] = *reinterpret_cast<const uint4*>(sa + lds_off<RB>(a_row0 + 16 * i, fh + 4));
            auto mm = [&](int i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 c = acc[i][j];
                    if constexpr (F16S) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&f1b[j]), *reinterpret_cast<f16x8*>(&f0a[i]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&f0b[j]), *reinterpret_cast<f16x8*>(&f1a[i]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&f0b[j]), *reinterpret_cast<f16x8*>(&f0a[i]), c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&f0b[j]), *reinterpret_cast<bf16x8*>(&f0a[i]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&f1b[j]), *reinterpret_cast<bf16x8*>(&f1a[i]), c, 0, 0, 0);
                    }
                    acc[i][j] = c;
                }
            };
#pragma unroll
            for (int i = 0; i < MT - SPLIT; ++i) mm(i);
            __builtin_amdgcn_sched_barrier(0);  // MFMAs touch no memory: without these hipcc moves them across the fence at will
            dma_fence();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = MT - SPLIT; i < MT; ++i) mm(i);
            continue;
        }
        uint4 ha[(F16S || FP8) ? MT : 1], hb[(F16S || FP8) ? 4 : 1];  // first 16-byte chunk of a fragment, kept until the second is read
        (void)ha; (void)hb;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            uint4 fa[MT], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[j] = *reinterpret_cast<const uint4*>(sb + lds_off<RB>(b_row0 + 4 * j, fh + 4 * g));
#pragma unroll
            for (int i = 0; i < MT; ++i)
                fa[i] = *reinterpret_cast<const uint4*>(sa + lds_off<RB>(a_row0 + 16 * i, fh + 4 * g));
            if constexpr (F16S) {
                // chunk g = 0 holds the hi halves of this 32-element slice, g = 1 the lo halves: keep the hi
                // fragments and fold the three products in once both are in registers (below)
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) ha[i] = fa[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) hb[j] = fb[j];
                } else {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 c = acc[i][j];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&fb[j]),
                                                                       *reinterpret_cast<f16x8*>(&ha[i]), c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&hb[j]),
                                                                       *reinterpret_cast<f16x8*>(&fa[i]), c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<f16x8*>(&hb[j]),
                                                                       *reinterpret_cast<f16x8*>(&ha[i]), c, 0, 0, 0);
                            acc[i][j] = c;
                        }
                }
            } else if constexpr (BF16) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<bf16x8*>(&fb[j]), *reinterpret_cast<bf16x8*>(&fa[i]), acc[i][j], 0, 0, 0);
            } else if constexpr (FP8) {
                // Block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3): K = 128 per instruction at twice the
                // bf16 rate per clock — the non-scaled fp8 MFMA runs at the bf16 rate.  Lane (fr, fh) supplies 32 bytes of
                // its row: the two 16-byte chunks fh and fh + 4 of the 128-byte slice (the k order inside the slice is
                // the same permutation for both operands; layout probed on hardware, tools/probes/mx_mfma_layout_probe.hip).
                // Scales: E8M0 bytes, 0x7f = 2^0 for every 32-block (the power-of-two tensor scales live in `alpha`).
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) ha[i] = fa[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) hb[j] = fb[j];
                } else {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            i32x8 wa, xa;
                            wa[0] = hb[j].x; wa[1] = hb[j].y; wa[2] = hb[j].z; wa[3] = hb[j].w;
                            wa[4] = fb[j].x; wa[5] = fb[j].y; wa[6] = fb[j].z; wa[7] = fb[j].w;
                            xa[0] = ha[i].x; xa[1] = ha[i].y; xa[2] = ha[i].z; xa[3] = ha[i].w;
                            xa[4] = fa[i].x; xa[5] = fa[i].y; xa[6] = fa[i].z; xa[7] = fa[i].w;
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                                         0x7f7f7f7f);
                        }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const float av = __uint_as_float(reinterpret_cast<const unsigned*>(&fa[i])[e]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float bv = __uint_as_float(reinterpret_cast<const unsigned*>(&fb[j])[e]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
                        }
                    }
            }
        }
        dma_fence();
    }
    // Persistent walk: the first K slice of this workgroup's next tile is put in flight before the epilogue, so
    // its latency (and the next workgroup launch) hides under the GELU / store tail.  Stage 0 is free: every
    // wave passed the fence that closed the last slice.
    const int bm_done = bm, bn_done = bn;
    tl += gw;
    const bool more = tl < chunk_size;
    if (more) {
        setup_tile(tl);
        stage_slice(0, 0);
    }
#ifdef SWC_GEMM_STAMP
    const unsigned long long st_e0 = stamp();
#endif
    // the other stage buffer is free until the fence below (the next tile's first slice goes to stage 0): per-wave scratch
    // (not for 256-row tiles: no register to spare, their K loops spilled or lost their schedule)
    gemm_epilogue<OutT, MT, MT < 8, ACT>(p, acc, bm_done, bn_done, wr, wc, fr, fh,
                            reinterpret_cast<float*>(smem + STAGE_BYTES + wave_u * EPI_WAVE_BYTES),
                            p.nt_mode == 2 || (p.nt_mode == 1 && !more));
#ifdef SWC_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_e1 = stamp();
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* o = g_stamps + ((long)blockIdx.x * 8 + wave) * 8;
        o[0] += st_cmp; o[1] += st_vm; o[2] += st_bar; o[3] += st_n; o[4] += st_e1 - st_e0; o[5] += 1; o[6] += st_e0 - st_last;
    }
    st_cmp = st_vm = st_bar = st_n = 0; st_last = 0;
#endif
    if (!more) break;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dma_fence();
    }
}

template <int MODE, typename OutT, int MT, int WM, int WN, bool PLAIN, int RB = 128, int ACT = 2>
int launch_k(GemmP p, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * 64;
    constexpr int LDS = 2 * (BM + BN) * RB;
    {
        constexpr int BK = RB / (MODE == SWC_BF16 ? 2 : (MODE == SWC_FP8 ? 1 : 4));
        p.kc_per_tap = (p.K + BK - 1) / BK;
    }
    p.n_tiles_n = (p.N + BN - 1) / BN;
    p.n_tiles_m = (p.M + BM - 1) / BM;
    {
        const int forced = tuning_env("SWC_GEMM_BAND");
        p.band = forced > 0 ? forced : (p.n_tiles_n >= 12 ? 4 : 1);
    }
    const long nwg = (long)p.n_tiles_n * p.n_tiles_m;
    if (nwg >= (1L << 30)) {
        swc_set_error("swc_gemm: grid too large");
        return SWC_E_ARG;
    }
    auto kern = gemm_kernel<MODE, OutT, MT, WM, WN, PLAIN, RB, ACT>;
    if (LDS > 64 * 1024) SWC_ENABLE_LDS(kern, LDS, "swc_gemm");
    // persistent grid: one workgroup per CU for the 8-wave geometries, two for the 4-wave one (256 CUs)
    const int persist = tuning_env("SWC_GEMM_NOPERSIST") ? 0 : 1;
    const long slots = 256L * (WM * WN == 4 ? 2 : 1);
    const long grid = (persist && nwg > slots) ? slots : nwg;
    // Non-temporal epilogue stores (profiles/r03_gemm_store_policy.txt): alone on the chip the split-f16 fc2 gains 12 % from
    // them (one tile per workgroup, whole 256-byte row pieces behind a long K loop), every other shape loses 0 - 40 %; inside
    // the pipeline the same fc2 / out-proj launches run 1 % SLOWER with them (129 -> 130 us, same-box rocprof A/B): off.
    // The mode stays selectable in tuning builds.
    p.nt_mode = tuning_env("SWC_GEMM_NT", 0);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), LDS, s, p);
    return SWC_OK;
}

template <int MODE, typename OutT, int MT, int WM, int WN, int RB = 128>
int launch(GemmP p, hipStream_t s) {
    constexpr int BK = RB / (MODE == SWC_BF16 ? 2 : (MODE == SWC_FP8 ? 1 : 4));
    const long es = MODE == SWC_BF16 ? 2 : (MODE == SWC_FP8 ? 1 : 4);
    const bool plain = (p.taps == 1) && (p.K % BK == 0) && (p.stride == 1) && (p.pad == 0) && (p.t_in == p.t_out) &&
                       p.lda * es < (1L << 24) && p.ldw * es < (1L << 24);  // 24-bit row pitch: offsets by v_mul_u32_u24
    // plain GEMMs with 16- / 8-bit outputs (qkv, fc1 + GELU, pwconv1 + GELU): the activation is compiled in, one epilogue
    // body per kernel
    if constexpr (sizeof(OutT) < 4 && MODE != SWC_F32) {
        if (plain) return p.act == SWC_ACT_GELU ? launch_k<MODE, OutT, MT, WM, WN, true, RB, 1>(p, s) : launch_k<MODE, OutT, MT, WM, WN, true, RB, 0>(p, s);
        return launch_k<MODE, OutT, MT, WM, WN, false, RB, 2>(p, s);
    } else {
        return plain ? launch_k<MODE, OutT, MT, WM, WN, true, RB>(p, s) : launch_k<MODE, OutT, MT, WM, WN, false, RB>(p, s);
    }
}

}  // namespace

// SWC_GEMM_TILE=128|256 overrides the geometry choice (tuning builds only).
static int tile_override() { return tuning_env("SWC_GEMM_TILE"); }

extern "C" int swc_gemm(const swc_gemm_args* a, void* stream) {
    SWC_CHECK_ARG(a != nullptr, "swc_gemm: null args");
    SWC_CHECK_ARG(a->A && a->W && a->C, "swc_gemm: null operand");
    SWC_CHECK_ARG(a->M >= 0 && a->N > 0 && a->K > 0, "swc_gemm: bad M/N/K %d %d %d", a->M, a->N, a->K);
    SWC_CHECK_ARG(a->a_dtype == SWC_F32 || a->a_dtype == SWC_BF16 || a->a_dtype == SWC_F16S || a->a_dtype == SWC_FP8,
                  "swc_gemm: bad a_dtype");
    SWC_CHECK_ARG(a->c_dtype == SWC_F32 || a->c_dtype == SWC_BF16 || a->c_dtype == SWC_F16S ||
                      (a->c_dtype == SWC_FP8 && a->a_dtype == SWC_FP8),
                  "swc_gemm: bad c_dtype (FP8 outputs come from FP8 operands only)");
    SWC_CHECK_ARG(a->c_dtype != SWC_F16S || a->a_dtype != SWC_FP8, "swc_gemm: fp8 operands have no split-f16 output");
    SWC_CHECK_ARG(a->c_dtype != SWC_F16S || a->a_dtype != SWC_BF16, "swc_gemm: bf16 operands have no split-f16 output");
    SWC_CHECK_ARG(a->c_dtype != SWC_BF16 || a->a_dtype != SWC_F16S, "swc_gemm: split-f16 operands have no bf16 output");
    SWC_CHECK_ARG(a->c_dtype != SWC_F16S || (a->N % 32 == 0 && a->ldc % 32 == 0 && aligned16(a->C)),
                  "swc_gemm: split-f16 output needs N, ldc multiples of 32 (N=%d)", a->N);
    SWC_CHECK_ARG(a->act == SWC_ACT_NONE || a->act == SWC_ACT_GELU, "swc_gemm: bad act");
    SWC_CHECK_ARG(!a->gamma || a->c_dtype == SWC_F32, "swc_gemm: gamma (per-column output scale) exists for f32 outputs only");
    const bool bf = a->a_dtype == SWC_BF16;
    const bool fs = a->a_dtype == SWC_F16S;
    const bool f8 = a->a_dtype == SWC_FP8;
    const int epc = bf ? 8 : (fs ? 32 : (f8 ? 16 : 4));
    SWC_CHECK_ARG(a->K % epc == 0, "swc_gemm: K=%d not a multiple of %d", a->K, epc);
    SWC_CHECK_ARG(a->lda % epc == 0 && a->ldw % epc == 0, "swc_gemm: lda/ldw break 16-byte rows");
    SWC_CHECK_ARG(aligned16(a->A) && aligned16(a->W), "swc_gemm: A/W not 16-byte aligned");
    SWC_CHECK_ARG(a->taps >= 1 && a->stride >= 1 && a->dil >= 1 && a->t_in >= 1 && a->t_out >= 1,
                  "swc_gemm: bad conv geometry");
    SWC_CHECK_ARG(a->M % a->t_out == 0, "swc_gemm: M=%d not a multiple of t_out=%d", a->M, a->t_out);
    SWC_CHECK_ARG(a->ldw >= (int64_t)a->taps * a->K, "swc_gemm: ldw < taps*K");
    SWC_CHECK_ARG(a->lda >= a->K && a->ldc >= a->N, "swc_gemm: lda < K or ldc < N");
    SWC_CHECK_ARG(!a->residual || a->ldr >= a->N, "swc_gemm: ldr < N");
    if (a->M == 0) return SWC_OK;

    GemmP p;
    p.A = (const char*)a->A;
    p.W = (const char*)a->W;
    p.C = a->C;
    p.bias = a->bias;
    p.gamma = a->gamma;
    p.residual = a->residual;
    p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldr = a->ldr;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.taps = a->taps; p.dil = a->dil; p.stride = a->stride; p.pad = a->pad;
    p.t_in = a->t_in; p.t_out = a->t_out;
    p.act = a->act;
    p.alpha = a->alpha == 0.0f ? 1.0f : a->alpha;
    p.out_scale = a->out_scale == 0.0f ? 1.0f : a->out_scale;
    const int bk = bf ? 64 : (f8 ? 128 : 32);
    p.kc_per_tap = (a->K + bk - 1) / bk;
    p.n_tiles_n = p.n_tiles_m = 0;
    p.nt_mode = -1;  // chosen per geometry in launch_k (tuning builds: SWC_GEMM_NT forces 0 / 1 / 2)
    p.stagger = tuning_env("SWC_GEMM_STAGGER", 0);
    p.sat = swc_sat_counter();
    hipStream_t s = (hipStream_t)stream;
    // geometry: the 256x256 / 8-wave tile pays off when its grid still fills the 256 CUs
    const long big_tiles = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
    bool big = (bf || f8) && a->N >= 256 && big_tiles >= 96;
    if (tile_override() == 128) big = false;
    if (tile_override() == 256) big = bf || f8;
    int rc;
    const int cd = a->c_dtype;
#define SWC_LAUNCH(MODE, MT, WM, WN)                                                        \
    (cd == SWC_BF16 ? launch<MODE, bf16_t, MT, WM, WN>(p, s)                                 \
                    : (cd == SWC_F16S ? launch<MODE, f16s_t, MT, WM, WN>(p, s) : launch<MODE, float, MT, WM, WN>(p, s)))
    // 16-bit outputs come in the operands' own format only (checked above): no split-f16 -> bf16 or bf16 -> split-f16 kernels
#define SWC_LAUNCH_B(MT, WM, WN) (cd == SWC_BF16 ? launch<SWC_BF16, bf16_t, MT, WM, WN>(p, s) : launch<SWC_BF16, float, MT, WM, WN>(p, s))
#define SWC_LAUNCH_S(MT, WM, WN) (cd == SWC_F16S ? launch<SWC_F16S, f16s_t, MT, WM, WN>(p, s) : launch<SWC_F16S, float, MT, WM, WN>(p, s))
    // 8-wave geometries differ only in the rows per tile (256 / 192 / 128 x 256 columns): pick the one whose grid
    // quantises best over the 256 CUs.  cost ~ rounds x (rows + fixed per-tile overhead); e.g. M = 16000, N = 768:
    // 189 tiles of 256 rows leave a quarter of the chip idle, 252 tiles of 192 rows fill it in one round.
    int mt = 8;
    if (((bf || f8) && big) || (fs && a->N >= 256 && big_tiles >= 96)) {
        const int forced = tuning_env("SWC_GEMM_MT");
        const long ntn = (a->N + 255) / 256;
        double best = 1e30;
        for (int cand : {8, 6, 4}) {
            const long tiles = ((a->M + cand * 32 - 1) / (cand * 32)) * ntn;
            const double cost = (double)((tiles + 255) / 256) * (cand + 1.5);
            if (cost < best - 1e-9) { best = cost; mt = cand; }
        }
        if (forced == 8 || forced == 6 || forced == 4) mt = forced;
    }
    // small grids: when the 128 x 128 tiling leaves CU slots empty (4-wave workgroups, two per CU), 64-row tiles double
    // the workgroup count (down- / up-sampler k7 convs: 6048 x 512 outputs = 192 tiles on 512 slots)
    const int small_env = tuning_env("SWC_GEMM_SMALL", 1);
    const long tiles128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128);
    const bool half_rows = small_env && tiles128 < 384 && a->M >= 512;
    if (f8) {
#define SWC_LAUNCH8(MT, WM, WN)                                                   \
    (cd == SWC_BF16 ? launch<SWC_FP8, bf16_t, MT, WM, WN>(p, s)                     \
                    : (cd == SWC_FP8 ? launch<SWC_FP8, fp8_t, MT, WM, WN>(p, s) : launch<SWC_FP8, float, MT, WM, WN>(p, s)))
        if (big)
            rc = mt == 8 ? SWC_LAUNCH8(8, 2, 4) : (mt == 6 ? SWC_LAUNCH8(6, 2, 4) : SWC_LAUNCH8(4, 2, 4));
        else
            rc = SWC_LAUNCH8(4, 2, 2);
#undef SWC_LAUNCH8
    } else if (bf) {
        if (big)
            rc = mt == 8 ? SWC_LAUNCH_B(8, 2, 4) : (mt == 6 ? SWC_LAUNCH_B(6, 2, 4) : SWC_LAUNCH_B(4, 2, 4));
        else
            rc = half_rows ? SWC_LAUNCH_B(2, 2, 2) : SWC_LAUNCH_B(4, 2, 2);
    } else if (fs) {
        bool bigs = a->N >= 256 && big_tiles >= 96;
        if (tile_override() == 128) bigs = false;
        if (tile_override() == 256) bigs = true;
        if (bigs)
            rc = mt == 8 ? SWC_LAUNCH_S(8, 2, 4) : (mt == 6 ? SWC_LAUNCH_S(6, 2, 4) : SWC_LAUNCH_S(4, 2, 4));
        else
            rc = half_rows ? SWC_LAUNCH_S(2, 2, 2) : SWC_LAUNCH_S(4, 2, 2);
    } else {
        rc = SWC_LAUNCH(SWC_F32, 4, 2, 2);
    }
#undef SWC_LAUNCH
#undef SWC_LAUNCH_B
#undef SWC_LAUNCH_S
    if (rc != SWC_OK) return rc;
    SWC_CHECK_LAUNCH("swc_gemm");
    return SWC_OK;
}

#ifdef SWC_GEMM_STAMP
extern "C" int swc_debug_stamps(void* host_out, int reset) {
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 512 * 8 * 8) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zeros[512 * 8 * 8];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

