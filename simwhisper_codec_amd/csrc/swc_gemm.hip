// swc_gemm: C = epi(A (*) W^T) on gfx950 MFMA, plain GEMM or implicit-GEMM Conv1d
// over frame-major activations.  One 128x128 output tile per 256-thread workgroup
// (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles), K walked in 128-byte
// slices (32 f32 / 64 bf16) through a double-buffered, XOR-swizzled LDS image.
//
// f32 : v_mfma_f32_16x16x4_f32  (exact f32 fma chain, the parity path)
// bf16: v_mfma_f32_16x16x32_bf16 (f32 accumulate)
//
// Fragment addressing (both dtypes): lane l, r = l & 15, h = l >> 4 reads the 16-byte
// chunk c = h + 4g (g = 0,1) of LDS row r.  For f32 the chunk's 4 floats are 4
// successive MFMA k-steps (element j of every lane is k = 16g + 4h + j: a permuted but
// complete walk of the slice); for bf16 the chunk is the 8-element operand of one
// 16x16x32 step (k = 32g + 8h + j).
#include "swc_common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int ROW_BYTES = 128;            // bytes of K per LDS row per slice
constexpr int TILE_BYTES = BM * ROW_BYTES;  // 16 KiB per operand per stage
constexpr int NTHREADS = 256;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// 16 bytes per lane, global -> LDS without a VGPR round trip.  `lds_base` must be wave-uniform:
// the hardware writes lane l at lds_base + 16 * l.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return ((row >> 1) ^ ((row >> 4) << 1)) & 7; }

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // conflict-free for the 16-lane groups of ds_read_b128, both for 16 consecutive rows (activation
    // fragments) and for the permuted weight rows {16a + 4j + b} of the transposed-product epilogue
    // (checked exhaustively over the four lane groups): rows r, r+2 share a 256-byte bank row.
    return row * ROW_BYTES + ((chunk ^ swz(row)) << 4);
}

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
    int kc_per_tap;  // ceil(K / BK)
    int n_tiles_n, n_tiles_m;
};

template <bool BF16, typename OutT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmP p) {
    constexpr int ES = BF16 ? 2 : 4;          // element size
    constexpr int EPC = 16 / ES;              // elements per 16-byte chunk
    constexpr int BK = ROW_BYTES / ES;        // elements of K per slice

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [stage][A|B][TILE_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so
    // give each XCD a contiguous run of tiles (neighbours share the A panel in its L2).
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / p.n_tiles_n;
    const int tn = bid - tm * p.n_tiles_n;
    const int bm = tm * BM, bn = tn * BN;

    // ---- per-thread staging geometry: 4 A chunks + 4 W chunks per slice
    const int ld_chunk = tid & 7;
    int a_b[4], a_t[4];
    bool a_rowok[4];
    long w_rowoff[4];
    bool w_rowok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int r = bm + row;
        a_rowok[i] = r < p.M;
        const int rr = a_rowok[i] ? r : 0;
        a_b[i] = rr / p.t_out;
        a_t[i] = rr - a_b[i] * p.t_out;
        const int n = bn + row;
        w_rowok[i] = n < p.N;
        w_rowoff[i] = (long)(w_rowok[i] ? n : 0) * p.ldw;
    }

    // Direct global->LDS staging (global_load_lds_dwordx4): one wave-instruction fills 8 LDS rows
    // (1 KiB, lane-linear), so the XOR swizzle lives on the SOURCE address: the lane that lands on
    // chunk position p of row r fetches logical chunk p ^ ((r >> 1) & 7).  Lanes whose row / tap /
    // K-chunk is out of range fetch a 16-byte device zero page instead (no zero-fill path exists).
    int a_chunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        a_chunk[i] = ld_chunk ^ swz(row);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage_slice = [&](int kt, int stage) {
        const int tap = kt / p.kc_per_tap;
        const int kc = kt - tap * p.kc_per_tap;
        const int shift = tap * p.dil - p.pad;
        char* sa = smem + stage * 2 * TILE_BYTES;
        char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k0 = kc * BK + a_chunk[i] * EPC;
            const bool kok = k0 < p.K;
            const int ts = a_t[i] * p.stride + shift;
            const bool ok = a_rowok[i] && kok && ts >= 0 && ts < p.t_in;
            const char* src = reinterpret_cast<const char*>(g_zero16);
            if (ok) src = p.A + (((long)a_b[i] * p.t_in + ts) * p.lda + k0) * ES;
            glds16(src, sa + (32 * i + 8 * wave_u) * ROW_BYTES);
            const char* srcw = reinterpret_cast<const char*>(g_zero16);
            if (w_rowok[i] && kok) srcw = p.W + (w_rowoff[i] + (long)tap * p.K + k0) * ES;
            glds16(srcw, sb + (32 * i + 8 * wave_u) * ROW_BYTES);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nkt = p.taps * p.kc_per_tap;
    stage_slice(0, 0);
    __syncthreads();  // drains vmcnt (the LDS-DMA) before the barrier

    const int fr = lane & 15, fh = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage_slice(kt + 1, cur ^ 1);
        const char* sa = smem + cur * 2 * TILE_BYTES;
        const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra_row = wr * 64 + i * 16 + fr;
                fa[i] = *reinterpret_cast<const uint4*>(sa + lds_off(ra_row, fh + 4 * g));
                // weight rows are taken in the order {16a + 4i + b}: with the weight fragment as the
                // MFMA's row operand, lane (fr, fh) then owns 16 CONTIGUOUS output columns 16fh + 4i + e
                const int rb_row = wc * 64 + 16 * (fr >> 2) + 4 * i + (fr & 3);
                fb[i] = *reinterpret_cast<const uint4*>(sb + lds_off(rb_row, fh + 4 * g));
            }
            if constexpr (BF16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<bf16x8*>(&fb[j]), *reinterpret_cast<bf16x8*>(&fa[i]),
                            acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float av = __uint_as_float(reinterpret_cast<const unsigned*>(&fa[i])[e]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float bv =
                                __uint_as_float(reinterpret_cast<const unsigned*>(&fb[j])[e]);
                            acc[i][j] =
                                __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  Transposed product: D[n][m], so lane (fr, fh) holds, for row m = 16i + fr,
    // the 16 contiguous columns 16fh + 4j + e (j = 0..3, e = 0..3) of its wave's 64-column slab.
    OutT* C = reinterpret_cast<OutT*>(p.C);
    const int col0 = bn + wc * 64 + 16 * fh;
    const bool vec_ok = (col0 + 16 <= p.N) && ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.gamma) & 15) == 0);
    float bv[16], gv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int col = col0 + c;
        bv[c] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        gv[c] = (p.gamma && col < p.N) ? p.gamma[col] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = bm + wr * 64 + i * 16 + fr;
        if (row >= p.M) continue;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[i][j][e] + bv[4 * j + e];
                if (p.act == SWC_ACT_GELU) x = gelu_erf(x);
                v[4 * j + e] = x * gv[4 * j + e];
            }
        if (vec_ok) {
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
                    reinterpret_cast<float4*>(cp)[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint4 u;
                    u.x = (unsigned)f32_to_bf16(v[8 * j + 0]) | ((unsigned)f32_to_bf16(v[8 * j + 1]) << 16);
                    u.y = (unsigned)f32_to_bf16(v[8 * j + 2]) | ((unsigned)f32_to_bf16(v[8 * j + 3]) << 16);
                    u.z = (unsigned)f32_to_bf16(v[8 * j + 4]) | ((unsigned)f32_to_bf16(v[8 * j + 5]) << 16);
                    u.w = (unsigned)f32_to_bf16(v[8 * j + 6]) | ((unsigned)f32_to_bf16(v[8 * j + 7]) << 16);
                    reinterpret_cast<uint4*>(cp)[j] = u;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int col = col0 + c;
                if (col >= p.N) continue;
                float x = v[c];
                if (p.residual) x += p.residual[(long)row * p.ldr + col];
                store_out<OutT>(C + (long)row * p.ldc + col, x);
            }
        }
    }
}

}  // namespace

extern "C" int swc_gemm(const swc_gemm_args* a, void* stream) {
    SWC_CHECK_ARG(a != nullptr, "swc_gemm: null args");
    SWC_CHECK_ARG(a->A && a->W && a->C, "swc_gemm: null operand");
    SWC_CHECK_ARG(a->M >= 0 && a->N > 0 && a->K > 0, "swc_gemm: bad M/N/K %d %d %d", a->M, a->N, a->K);
    SWC_CHECK_ARG(a->a_dtype == SWC_F32 || a->a_dtype == SWC_BF16, "swc_gemm: bad a_dtype");
    SWC_CHECK_ARG(a->c_dtype == SWC_F32 || a->c_dtype == SWC_BF16, "swc_gemm: bad c_dtype");
    SWC_CHECK_ARG(a->act == SWC_ACT_NONE || a->act == SWC_ACT_GELU, "swc_gemm: bad act");
    const bool bf = a->a_dtype == SWC_BF16;
    const int epc = bf ? 8 : 4;
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
    const int bk = bf ? 64 : 32;
    p.kc_per_tap = (a->K + bk - 1) / bk;
    p.n_tiles_n = (a->N + BN - 1) / BN;
    p.n_tiles_m = (a->M + BM - 1) / BM;
    const long nwg = (long)p.n_tiles_n * p.n_tiles_m;
    SWC_CHECK_ARG(nwg < (1L << 30), "swc_gemm: grid too large");
    const size_t lds = 4 * TILE_BYTES;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)nwg), block(NTHREADS);
    if (bf) {
        if (a->c_dtype == SWC_BF16)
            hipLaunchKernelGGL((gemm_kernel<true, bf16_t>), grid, block, lds, s, p);
        else
            hipLaunchKernelGGL((gemm_kernel<true, float>), grid, block, lds, s, p);
    } else {
        if (a->c_dtype == SWC_BF16)
            hipLaunchKernelGGL((gemm_kernel<false, bf16_t>), grid, block, lds, s, p);
        else
            hipLaunchKernelGGL((gemm_kernel<false, float>), grid, block, lds, s, p);
    }
    SWC_CHECK_LAUNCH("swc_gemm");
    return SWC_OK;
}
