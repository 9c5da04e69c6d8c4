// HBM-bound kernels of the hot path: LayerNorm, depthwise-k7+LayerNorm, anti-aliased
// SnakeBeta, FSQ encode/decode, the log-mel pieces around the DFT/mel GEMMs, the
// ConvTranspose1d col2im tail and the ISTFT spectrum + overlap-add.  All operate on
// frame-major [B][T][C] activations so that a wave reads whole rows (C contiguous).
#include "swc_common.h"

namespace {

template <typename OutT>
__device__ __forceinline__ void store4(OutT* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    uint2 u;
    u.x = bf16_pack2(a, b);
    u.y = bf16_pack2(c, d);
    *reinterpret_cast<uint2*>(p) = u;
}

// store 4 consecutive columns col..col+3 of a row; `row` points at the row start in OutT units
// (split-f16 rows are addressed in halves: 2 per logical column)
template <typename OutT>
__device__ __forceinline__ void store_row4(OutT* row, int col, float a, float b, float c, float d, float& amax) {
    if constexpr (__is_same(OutT, fp8_t)) {
        *reinterpret_cast<unsigned*>(row + col) = fp8_pack4(a * SWC_FP8_ACT_SCALE, b * SWC_FP8_ACT_SCALE,
                                                            c * SWC_FP8_ACT_SCALE, d * SWC_FP8_ACT_SCALE, amax);
    } else if constexpr (__is_same(OutT, f16s_t)) {
        f16s_store4(reinterpret_cast<unsigned short*>(row), col, a * SWC_F16S_ACT_SCALE, b * SWC_F16S_ACT_SCALE,
                    c * SWC_F16S_ACT_SCALE, d * SWC_F16S_ACT_SCALE, amax);
    } else {
        store4<OutT>(row + col, a, b, c, d);
    }
}
template <typename OutT>
__device__ __forceinline__ long row_units(long C) { return __is_same(OutT, f16s_t) ? 2 * C : C; }
// range of the output format of a producer (infinite for f32 / bf16) and its slot in the saturation counter pair
template <typename OutT>
__device__ __forceinline__ void sat_commit_out(unsigned* sat, float amax) {
    if constexpr (__is_same(OutT, f16s_t)) sat_commit(sat, 0, amax, SWC_F16S_LIMIT);
    if constexpr (__is_same(OutT, fp8_t)) sat_commit(sat, 1, amax, SWC_FP8_LIMIT);
}

// ------------------------------------------------------------------ LayerNorm
constexpr int LN_MAXV = 8;  // float4 per lane => C <= 2048

template <typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bia,
                                                        const int* __restrict__ lens, long rows, int tw,
                                                        int t_in, int t_out, int C, float eps, unsigned* sat,
                                                        const int* __restrict__ row_start) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int b = (int)(r / tw), t = (int)(r - (long)b * tw);
    const float* xr = x + ((row_start ? (long)row_start[b] : (long)b * t_in) + t) * C;
    OutT* yr = y + ((long)b * t_out + t) * row_units<OutT>(C);
    const int nv = C >> 2;  // float4 count
    if (t >= t_in || (lens && t >= lens[b])) {  // masked rows, and the zero extension beyond the input (t_out > t_in)
        float z_ = 0.f;
        for (int i = lane; i < nv; i += 64) store_row4<OutT>(yr, 4 * i, 0.f, 0.f, 0.f, 0.f, z_);
        return;
    }
    float amax = 0.f;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nv) {
            v[k] = *reinterpret_cast<const float4*>(xr + 4 * i);
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
    const float mean = wave_sum_dpp(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            const float a = v[k].x - mean, bb = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + bb * bb) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            const float4 ww = *reinterpret_cast<const float4*>(w + 4 * i);
            const float4 bb = *reinterpret_cast<const float4*>(bia + 4 * i);
            store_row4<OutT>(yr, 4 * i, (v[k].x - mean) * rstd * ww.x + bb.x, (v[k].y - mean) * rstd * ww.y + bb.y,
                             (v[k].z - mean) * rstd * ww.z + bb.z, (v[k].w - mean) * rstd * ww.w + bb.w, amax);
        }
    }
    sat_commit_out<OutT>(sat, amax);
}

// RPW rows per wave, C == 256 * NV exactly (the path's 512 and 768): all rows' loads are in flight together, the
// reductions interleave, the affine parameters are read once per wave, and the grid has 1/RPW of the workgroups.
template <typename OutT, int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                         const float* __restrict__ w, const float* __restrict__ bia,
                                                         const int* __restrict__ lens, long rows, int tw, int t_in,
                                                         int t_out, float eps, unsigned* sat,
                                                         const int* __restrict__ row_start) {
    constexpr int C = 256 * NV;
    const int lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (r0 >= rows) return;
    const float* xr[RPW];
    OutT* yr[RPW];
    bool live[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const long r = r0 + u;
        const bool in = r < rows;
        const int b = in ? (int)(r / tw) : 0, t = in ? (int)(r - (long)b * tw) : 0;
        live[u] = in && t < t_in && !(lens && t >= lens[b]);
        // packed input (row_start != NULL): utterance b's rows start at row_start[b]; only live rows are dereferenced
        xr[u] = x + ((row_start && in ? (long)row_start[b] : (long)b * t_in) + t) * C;
        yr[u] = y + ((long)b * t_out + t) * row_units<OutT>(C);
        if (in && !live[u]) {  // masked rows and the zero extension beyond the input
            float z_ = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) store_row4<OutT>(yr[u], 4 * (lane + 64 * k), 0.f, 0.f, 0.f, 0.f, z_);
        }
    }
    float4 v[RPW][NV];
    float s[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) s[u] = 0.f;
#pragma unroll
    for (int u = 0; u < RPW; ++u)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            v[u][k] = live[u] ? *reinterpret_cast<const float4*>(xr[u] + 4 * (lane + 64 * k)) : make_float4(0.f, 0.f, 0.f, 0.f);
            s[u] += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
        }
    float4 ww[NV], bb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        ww[k] = *reinterpret_cast<const float4*>(w + 4 * (lane + 64 * k));
        bb[k] = *reinterpret_cast<const float4*>(bia + 4 * (lane + 64 * k));
    }
    float mean[RPW], q[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) q[u] = 0.f;
#pragma unroll
    for (int u = 0; u < RPW; ++u) mean[u] = wave_sum_dpp(s[u]) / (float)C;
#pragma unroll
    for (int u = 0; u < RPW; ++u)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float a = v[u][k].x - mean[u], b_ = v[u][k].y - mean[u], c = v[u][k].z - mean[u], d = v[u][k].w - mean[u];
            q[u] += (a * a + b_ * b_) + (c * c + d * d);
        }
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const float rstd = 1.0f / sqrtf(wave_sum_dpp(q[u]) / (float)C + eps);
        if (live[u]) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                store_row4<OutT>(yr[u], 4 * (lane + 64 * k), (v[u][k].x - mean[u]) * rstd * ww[k].x + bb[k].x,
                                 (v[u][k].y - mean[u]) * rstd * ww[k].y + bb[k].y, (v[u][k].z - mean[u]) * rstd * ww[k].z + bb[k].z,
                                 (v[u][k].w - mean[u]) * rstd * ww[k].w + bb[k].w, amax);
        }
    }
    sat_commit_out<OutT>(sat, amax);
}

// ------------------------------------------------- depthwise k7 conv + LayerNorm
// Strip kernel: a workgroup stages S+6 consecutive frames of one utterance in LDS, so every input row crosses
// the vector-memory path once (plus the 6-row halo) instead of 7 times, and keeps the 7xC taps, the bias and
// the LayerNorm affine in registers for the whole strip; each wave then turns out S/4 frames from LDS.
// HBM traffic per frame: C*4 B in (f32 residual stream) + C*sizeof(OutT) out.
constexpr int DW_MAXV = 4;  // float4 per lane: C <= 1024
constexpr int DW_FG = 4;    // frames a wave finishes together (their reductions overlap)

// rows t0-3 .. t0+S+2 of strip `sid` -> registers (row wv + 4r for r < NR); all loads in flight together
template <int NK, int S, bool FULL>
__device__ __forceinline__ void dw_fetch(float4 (&st)[(S + 6 + 3) / 4][NK], const float* __restrict__ x, int sid,
                                         int nst, int T, int nv, int wv, int lane) {
    const int b = sid / nst, t0 = (sid - b * nst) * S;
    const float4* xb = reinterpret_cast<const float4*>(x) + ((long)b * T + t0 - 3) * nv + lane;
#pragma unroll
    for (int r = 0; r < (S + 6 + 3) / 4; ++r) {
        const int rr = wv + 4 * r, ts = t0 + rr - 3;  // wave-uniform
        const bool in = rr < S + 6 && ts >= 0 && ts < T;  // zero padding per utterance (Conv1d padding=3)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (in && (FULL || lane + 64 * k < nv)) st[r][k] = xb[(long)rr * nv + 64 * k];
            else st[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// FULL: C == 1024 * NK / 4 exactly (every lane owns NK float4 of a row), no per-lane predicates
template <typename OutT, int NK, int S, bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void dwconv7_ln_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                       const float* __restrict__ w,     // [7][C]
                       const float* __restrict__ bias,  // [C]
                       const float* __restrict__ lw, const float* __restrict__ lb, int T, int C, int nst,
                       int nstrips, int per, float eps) {
    extern __shared__ float4 dw_sm[];  // [S + 6][C / 4]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nv = C >> 2;
    const float inv_c = 1.0f / (float)C;
    constexpr int NR = (S + 6 + 3) / 4;
    float4 st[NR][NK];
    // each workgroup walks `per` consecutive strips (the halo rows it re-reads were just loaded), and the
    // workgroups of one XCD (blockIdx % 8) own one contiguous range of strips, so neighbours share an L2
    const int per_xcd = gridDim.x >> 3;
    const int order = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    int sid = order * per;
    const int send = min(sid + per, nstrips);
    if (sid >= send) return;
    dw_fetch<NK, S, FULL>(st, x, sid, nst, T, nv, wv, lane);
    float4 wr[7][NK], br[NK], gw[NK], gb[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int i = lane + 64 * k;
        if (FULL || i < nv) {
#pragma unroll
            for (int j = 0; j < 7; ++j) wr[j][k] = reinterpret_cast<const float4*>(w + (long)j * C)[i];
            br[k] = reinterpret_cast<const float4*>(bias)[i];
            gw[k] = reinterpret_cast<const float4*>(lw)[i];
            gb[k] = reinterpret_cast<const float4*>(lb)[i];
        } else {
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 7; ++j) wr[j][k] = z4;
            br[k] = gw[k] = gb[k] = z4;
        }
    }
    constexpr int PER = S / 4;
    static_assert(PER % DW_FG == 0, "strip = 4 waves x groups of DW_FG frames");
    // persistent walk: while a strip is computed from LDS the next one is already in flight to registers
    for (;;) {
        const int b = sid / nst, t0 = (sid - b * nst) * S;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int rr = wv + 4 * r;
            if (rr < S + 6) {
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    if (FULL || lane + 64 * k < nv) dw_sm[rr * nv + lane + 64 * k] = st[r][k];
            }
        }
        __syncthreads();
        const int nxt = sid + 1;
        if (nxt < send) dw_fetch<NK, S, FULL>(st, x, nxt, nst, T, nv, wv, lane);
        for (int g = 0; g < PER / DW_FG; ++g) {
            const int f0 = wv * PER + g * DW_FG;
            if (t0 + f0 >= T) break;
            float4 v[DW_FG][NK];
            float sum[DW_FG], sq[DW_FG];
#pragma unroll
            for (int u = 0; u < DW_FG; ++u) {
#pragma unroll
                for (int k = 0; k < NK; ++k) v[u][k] = br[k];
            }
            // rows f0 .. f0+DW_FG+5, each read once; frame u takes row p as tap j = p - u (taps in order 0..6)
            const float4* rows = dw_sm + f0 * nv + lane;
#pragma unroll
            for (int pr = 0; pr < DW_FG + 6; ++pr) {
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (FULL || lane + 64 * k < nv) {
                        const float4 xv = rows[pr * nv + 64 * k];
#pragma unroll
                        for (int u = 0; u < DW_FG; ++u) {
                            const int j = pr - u;
                            if (j >= 0 && j < 7) {
                                v[u][k].x += xv.x * wr[j][k].x; v[u][k].y += xv.y * wr[j][k].y;
                                v[u][k].z += xv.z * wr[j][k].z; v[u][k].w += xv.w * wr[j][k].w;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < DW_FG; ++u) {
                sum[u] = 0.f;
#pragma unroll
                for (int k = 0; k < NK; ++k) sum[u] += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
            }
#pragma unroll
            for (int u = 0; u < DW_FG; ++u) sum[u] = wave_sum_dpp(sum[u]);
#pragma unroll
            for (int u = 0; u < DW_FG; ++u) {
                const float mean = sum[u] * inv_c;
                sq[u] = 0.f;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    v[u][k].x -= mean; v[u][k].y -= mean; v[u][k].z -= mean; v[u][k].w -= mean;
                    if (FULL || lane + 64 * k < nv)
                        sq[u] += (v[u][k].x * v[u][k].x + v[u][k].y * v[u][k].y) +
                                 (v[u][k].z * v[u][k].z + v[u][k].w * v[u][k].w);
                }
            }
#pragma unroll
            for (int u = 0; u < DW_FG; ++u) sq[u] = wave_sum_dpp(sq[u]);
            OutT* yr = y + ((long)b * T + t0 + f0) * C + 4 * lane;
#pragma unroll
            for (int u = 0; u < DW_FG; ++u) {
                if (t0 + f0 + u < T) {  // wave-uniform
                    const float rstd = rsqrtf(sq[u] * inv_c + eps);
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        if (FULL || lane + 64 * k < nv)
                            store4<OutT>(yr + (long)u * C + 256 * k, v[u][k].x * rstd * gw[k].x + gb[k].x,
                                         v[u][k].y * rstd * gw[k].y + gb[k].y, v[u][k].z * rstd * gw[k].z + gb[k].z,
                                         v[u][k].w * rstd * gw[k].w + gb[k].w);
                    }
                }
            }
        }
        if (nxt >= send) break;
        sid = nxt;
        __syncthreads();  // every wave is done with this strip's rows
    }
}

// ------------------------------------------------------ anti-aliased SnakeBeta
struct Filt12 { float f[12]; };

// sin(a)^2 to f32 accuracy (|error| < 1.5e-7, a few ulp of the f32 product sinf(a) * sinf(a)) in 16 VALU instructions,
// against ~50 for libm's sinf, which made the snake kernel sine-bound: a = q pi/2 + r with |r| <= pi/4 (three-constant
// Cody-Waite reduction, exact for |q| < 2^15), sin^2 r by its Taylor series in r^2 (six terms: truncation < 3e-9), and
// sin^2 a = sin^2 r for even q, 1 - sin^2 r for odd q — no quadrant-dependent polynomial, no sign logic.
__device__ __forceinline__ float sin2_f32(float a) {
    if (fabsf(a) > 8192.0f) {  // beyond the reduction's exact range (never on this path's activations): libm
        const float sn = sinf(a);
        return sn * sn;
    }
    const float q = rintf(a * 0.63661977236758134308f);
    float r = fmaf(-q, 1.5703125f, a);
    r = fmaf(-q, 4.837512969970703125e-4f, r);
    r = fmaf(-q, 7.54978995489188216e-8f, r);
    const float u = r * r;
    float p = fmaf(u, -4.2755787e-6f, 1.4109347e-4f);   // -2048 / 12!,  512 / 10!
    p = fmaf(p, u, -3.1746032e-3f);                      // -128 / 8!
    p = fmaf(p, u, 4.4444444e-2f);                       //  32 / 6!
    p = fmaf(p, u, -3.3333334e-1f);                      //  -8 / 4!
    p = fmaf(p, u, 1.0f);
    const float s2 = u * p;
    return ((int)q & 1) ? 1.0f - s2 : s2;
}
// SN_TS outputs per thread strip; 6 halo pairs are recomputed per strip (1.75x the activations at 8).  Longer strips do
// less arithmetic but measured SLOWER on these small tensors (16: 21.1 us against 18.2 us at 8 for (32, 189, 512)): with a
// few waves per SIMD the long dependent chains of the activation are not covered, so parallelism wins over the halo.
template <typename OutT, int SN_TS>
__global__ __launch_bounds__(256) void snake_aa_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, Filt12 F, int T,
                                                       int C, unsigned* sat) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int b = blockIdx.z;
    const int t0 = blockIdx.y * SN_TS;
    const float* xc = x + (long)b * T * C + c;
    OutT* yb = y + (long)b * T * row_units<OutT>(C);
    const float al = alpha[c];
    const float ib = 1.0f / (beta[c] + 1e-9f);
    const int T2 = 2 * T;
    auto X = [&](int t) -> float {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        return xc[(long)t * C];
    };
    // xw[k] = x[clamp(u - 3 + k)], k = 0..6 for the current up-sample pair u
    // up[2u]   = 2 * sum_{d=-3..2} x[u+d] f[5-2d];  up[2u+1] = 2 * sum_{d=-2..3} x[u+d] f[6-2d]
    auto act = [&](float v) -> float {
        // bf16 outputs (decode side of the bf16-class presets) take the hardware sine (v_sin_f32, |error| ~1e-6 on
        // these arguments); f32 / split-f16 outputs (the down-sampler feeds the FSQ rounding) take sin2_f32 below
        if constexpr (__is_same(OutT, bf16_t)) {
            const float sn = __sinf(v * al);
            return v + ib * (sn * sn);
        } else {
            return v + ib * sin2_f32(v * al);
        }
    };
    // a-window for output t: a[clamp(2t - 5 + k)], k = 0..11  => pairs u = t-3 .. t+3 (partially)
    // Keep a ring of activated pairs: A0[u], A1[u] for u in [t-3, t+3].
    float A0[7], A1[7];
    auto pair = [&](int u, float& e0, float& e1) {
        // replicate padding of the activated, up-sampled signal: a index clamped to [0, 2T-1]
        // is applied by the caller through u clamping of indices; here u is a real pair index.
        float xv[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) xv[k] = X(u - 3 + k);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int d = -3; d <= 2; ++d) s0 += xv[d + 3] * F.f[5 - 2 * d];
#pragma unroll
        for (int d = -2; d <= 3; ++d) s1 += xv[d + 3] * F.f[6 - 2 * d];
        e0 = act(2.0f * s0);
        e1 = act(2.0f * s1);
    };
    auto pair_clamped = [&](int u, float& e0, float& e1) {
        // a[m] for m outside [0, 2T-1] replicates a[0] / a[2T-1]
        if (u < 0) {
            float t0_, t1_;
            pair(0, t0_, t1_);
            e0 = t0_; e1 = t0_;
        } else if (u >= T) {
            float t0_, t1_;
            pair(T - 1, t0_, t1_);
            e0 = t1_; e1 = t1_;
        } else {
            pair(u, e0, e1);
        }
    };
    float amax = 0.f;
    auto emit = [&](int t, float o) {
        if constexpr (__is_same(OutT, f16s_t)) {
            unsigned short hi, lo;
            f16s_split(o * SWC_F16S_ACT_SCALE, hi, lo, amax);
            unsigned short* rp = reinterpret_cast<unsigned short*>(yb) + (long)t * 2 * C + f16s_col(c);
            rp[0] = hi;
            rp[32] = lo;
        } else {
            store_out<OutT>(yb + (long)t * C + c, o);
        }
    };
    // Interior strips (every up-sample pair t0 - 3 .. t0 + SN_TS + 2 exists): the strip's 20 inputs are requested at
    // once and everything else runs from registers.  The generic loop below asks for 7 inputs per pair, one pair after
    // the other: 9 dependent memory round trips per strip made the kernel latency-bound (11 - 14 % of the HBM roofline
    // with the hardware sine as with libm's), not sine-bound.
    if (t0 >= 3 && t0 + SN_TS + 2 <= T - 1) {  // block-uniform
        constexpr int NP = SN_TS + 6;           // pairs
        float xw[NP + 6];
#pragma unroll
        for (int k = 0; k < NP + 6; ++k) xw[k] = X(t0 - 6 + k);
        float P0[NP], P1[NP];
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int d = -3; d <= 2; ++d) s0 += xw[m + d + 3] * F.f[5 - 2 * d];
#pragma unroll
            for (int d = -2; d <= 3; ++d) s1 += xw[m + d + 3] * F.f[6 - 2 * d];
            P0[m] = act(2.0f * s0);
            P1[m] = act(2.0f * s1);
        }
#pragma unroll
        for (int tt = 0; tt < SN_TS; ++tt) {
            float o = 0.f;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int m = k + 1 + 2 * tt;
                o += ((m & 1) ? P1[m >> 1] : P0[m >> 1]) * F.f[k];
            }
            emit(t0 + tt, o);
        }
        sat_commit_out<OutT>(sat, amax);
        return;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) pair_clamped(t0 - 3 + k, A0[k], A1[k]);
    const int tend = (t0 + SN_TS < T) ? t0 + SN_TS : T;
    for (int t = t0; t < tend; ++t) {
        pair_clamped(t + 3, A0[6], A1[6]);
        // out[t] = sum_k a[2t-5+k] f[k]; 2t-5 = 2(t-3)+1 -> starts at A1[0]
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int m = k + 1;  // offset from a[2(t-3)]
            const float av = (m & 1) ? A1[m >> 1] : A0[m >> 1];
            o += av * F.f[k];
        }
        emit(t, o);
#pragma unroll
        for (int k = 0; k < 6; ++k) { A0[k] = A0[k + 1]; A1[k] = A1[k + 1]; }
    }
    sat_commit_out<OutT>(sat, amax);
    (void)T2;
}

// ----------------------------------------------------------------------- FSQ
struct FsqC { float scale[4], offset[4], shift[4]; int levels[4]; };

__global__ void fsq_encode_kernel(const float* __restrict__ z, long ldz, float* __restrict__ zq,
                                  int* __restrict__ codes, const int* __restrict__ lens, int B, int T,
                                  int t_pad, int G, FsqC K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * t_pad * G;
    if (i >= total) return;
    const int g = (int)(i % G);
    const long bt = i / G;
    const int t = (int)(bt % t_pad), b = (int)(bt / t_pad);
    const bool valid = t < T && t < lens[b];
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    int idx = 0;
    if (valid) {
        const float4 zv = *reinterpret_cast<const float4*>(z + ((long)b * T + t) * ldz + 4 * g);
        const float in[4] = {zv.x, zv.y, zv.z, zv.w};
        int base = 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int half = K.levels[d] / 2;
            const float th = (float)tanh((double)__fadd_rn(in[d], K.shift[d]));
            const float comp = __fsub_rn(__fmul_rn(K.scale[d], th), K.offset[d]);
            const float c = rintf(comp);  // half-to-even == torch.round
            q[d] = __fdiv_rn(c, (float)half);
            idx += ((int)c + half) * base;
            base *= K.levels[d];
        }
    }
    *reinterpret_cast<float4*>(zq + ((long)b * t_pad + t) * (4L * G) + 4 * g) = make_float4(q[0], q[1], q[2], q[3]);
    codes[((long)g * B + b) * t_pad + t] = idx;
}

__global__ void fsq_decode_kernel(const long long* __restrict__ codes, float* __restrict__ zq, long ldq,
                                  const int* __restrict__ lens, int B, int T, int G, FsqC K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ng = (int)(ldq / 4);  // groups incl. zero padding columns
    const long total = (long)B * T * ng;
    if (i >= total) return;
    const int g = (int)(i % ng);
    const long bt = i / ng;
    const int t = (int)(bt % T), b = (int)(bt / T);
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    if (g < G && t < lens[b]) {
        long long idx = codes[((long)g * B + b) * T + t];
        long long base = 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int half = K.levels[d] / 2;
            // torch's `//` and `%` (quantizer.py:214-216) round towards -inf: identical to C for valid codes, and for
            // out-of-range / negative ones the same wrapped level instead of C's truncation
            long long qd = idx / base;
            if (idx % base < 0) --qd;
            long long nn = qd % K.levels[d];
            if (nn < 0) nn += K.levels[d];
            q[d] = __fdiv_rn((float)(nn - half), (float)half);
            base *= K.levels[d];
        }
    }
    *reinterpret_cast<float4*>(zq + ((long)b * T + t) * ldq + 4 * g) = make_float4(q[0], q[1], q[2], q[3]);
}

// ------------------------------------------------------------------- log-mel
__global__ __launch_bounds__(256) void mel_frames_kernel(const float* __restrict__ wav, long ld_wav, const int* __restrict__ n,
                                                        int n_pad, float* __restrict__ frames, int T, int vec_ok) {
    // one thread per float4 of a frame (100 per frame, hop 160 and the 200-sample centre offset are multiples of 4, so an
    // interior quad is one aligned 16-byte load); reflect padding and the zero extension beyond the length go element-wise
    const int b = blockIdx.y;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)T * 100) return;
    const int t = (int)(q / 100), j = (int)(q - (long)t * 100);
    const int nb = n[b];
    const float* wb = wav + (long)b * ld_wav;
    const int s0 = t * 160 + 4 * j - 200;
    float4 v;
    if (vec_ok && s0 >= 0 && s0 + 3 < nb && s0 + 3 < n_pad) {
        v = *reinterpret_cast<const float4*>(wb + s0);
    } else {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int s = s0 + i;
            if (s < 0) s = -s;
            if (s >= n_pad) s = 2 * (n_pad - 1) - s;
            e[i] = s < nb ? wb[s] : 0.f;
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
    }
    reinterpret_cast<float4*>(frames + ((long)b * T + t) * 400)[j] = v;
}

__global__ void mel_power_kernel(const float* __restrict__ dft, long ld, float* __restrict__ pw, long ldp,
                                 long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ldp) return;
    const long r = i / ldp;
    const int k = (int)(i - r * ldp);
    float v = 0.f;
    if (k < 201) {
        const float re = dft[r * ld + k], im = dft[r * ld + 201 + k];
        const float m = sqrtf(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));  // |X| then ** 2
        v = __fmul_rn(m, m);
    }
    pw[i] = v;
}

__device__ __forceinline__ int f32_ordered(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_f32(int i) {
    return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff);
}

constexpr int LM_PER_THREAD = 16;
__global__ __launch_bounds__(256) void mel_logmax_kernel(float* __restrict__ mel, long ld, int* __restrict__ umax_ord,
                                                         int T, int n_mel) {
    // grid (ceil(T*n_mel / (256*16)), B): 4096 elements per workgroup, ONE atomic per workgroup
    // (every wave hammering one address per utterance ran 14x slower than the stream itself)
    __shared__ float wmax[4];
    const int b = blockIdx.y;
    const long total = (long)T * n_mel;
    const long base = (long)blockIdx.x * (256 * LM_PER_THREAD) + threadIdx.x;
    float v = -INFINITY;
#pragma unroll
    for (int k = 0; k < LM_PER_THREAD; ++k) {
        const long i = base + 256L * k;
        if (i < total) {
            const long t = i / n_mel;
            const int m = (int)(i - t * n_mel);
            float* p = mel + ((long)b * T + t) * ld + m;
            const float x = log10f(fmaxf(*p, 1e-10f));
            *p = x;
            v = fmaxf(v, x);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (v > -INFINITY) atomicMax(umax_ord + b, f32_ordered(v));
    }
}

__global__ void ordered_init_kernel(const float* __restrict__ src, int* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f32_ordered(src[i]);
}

template <typename OutT>
__global__ void mel_final_kernel(const float* __restrict__ mel, long ld, const int* __restrict__ umax_ord,
                                 OutT* __restrict__ out, long ldo, int T, int n_mel) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * ldo) return;
    const long t = i / ldo;
    const int m = (int)(i - t * ldo);
    float v = 0.f;
    if (m < n_mel) {
        const float mx = ordered_f32(umax_ord[b]);
        v = fmaxf(mel[((long)b * T + t) * ld + m], __fsub_rn(mx, 8.0f));
        v = __fdiv_rn(__fadd_rn(v, 4.0f), 4.0f);
    }
    store_out<OutT>(out + ((long)b * T + t) * ldo + m, v);
}

// ------------------------------------------------------------ deconv col2im
template <typename OutT>
__global__ void deconv_col2im_kernel(const float* __restrict__ y3, const float* __restrict__ bias,
                                     OutT* __restrict__ out, long ldo, int T, int C, int s, int t_out) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)t_out * ldo) return;
    const int to = (int)(i / ldo);
    const int c = (int)(i - (long)to * ldo);
    float v = 0.f;
    if (c < C) {
        v = bias[c];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = to - j;
            if (d >= 0 && d % s == 0) {
                const int ti = d / s;
                if (ti < T) v += y3[(((long)b * T + ti) * 3 + j) * C + c];
            }
        }
    }
    store_out<OutT>(out + ((long)b * t_out + to) * ldo + c, v);
}

// 4 channels per thread (C % 4 == 0, ldo % 4 == 0): float4 taps in, one 8 / 16-byte store out
template <typename OutT>
__global__ void deconv_col2im4_kernel(const float* __restrict__ y3, const float* __restrict__ bias,
                                      OutT* __restrict__ out, long ldo, int T, int C, int s, int t_out) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long q = ldo >> 2;
    if (i >= (long)t_out * q) return;
    const int to = (int)(i / q);
    const int c = (int)(i - (long)to * q) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        v = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = to - j;
            if (d >= 0 && d % s == 0) {
                const int ti = d / s;
                if (ti < T) {
                    const float4 t4 = *reinterpret_cast<const float4*>(y3 + (((long)b * T + ti) * 3 + j) * C + c);
                    v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
                }
            }
        }
    }
    store4<OutT>(out + ((long)b * t_out + to) * ldo + c, v.x, v.y, v.z, v.w);
}

// -------------------------------------------------------------------- ISTFT
// one thread per (row, bin): magnitude and phase are read once and feed both the real and the imaginary column
// (split-f16 output, the ISTFT head of the 16-bit decode presets since round 4: the halves swc_cast_f32_f16s would write for the same
// value at SWC_F16S_ACT_SCALE — |S| <= 100 by the clip below, nothing to saturate)
template <typename OutT>
__device__ __forceinline__ void spec_store(OutT* s, long r, long lds, int col, float v) {
    if constexpr (__is_same(OutT, f16s_t)) {
        unsigned short* row = reinterpret_cast<unsigned short*>(s) + r * lds * 2;
        unsigned short hi, lo;
        f16s_split(v * SWC_F16S_ACT_SCALE, hi, lo);
        row[f16s_col(col)] = hi;
        row[f16s_col(col) + 32] = lo;
    } else {
        store_out<OutT>(s + r * lds + col, v);
    }
}
template <typename OutT>
__global__ void istft_spec2_kernel(const float* __restrict__ h, long ldh, OutT* __restrict__ s, long lds, long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 321 + (int)(lds - 642);  // bins + the zero columns beyond 642
    if (i >= rows * per) return;
    const long r = i / per;
    const int k = (int)(i - r * per);
    if (k >= 321) {
        spec_store<OutT>(s, r, lds, 642 + (k - 321), 0.f);
        return;
    }
    float mag = expf(h[r * ldh + k]);
    mag = fminf(mag, 100.0f);
    const float ph = h[r * ldh + 321 + k];
    spec_store<OutT>(s, r, lds, k, mag * cosf(ph));
    spec_store<OutT>(s, r, lds, 321 + k, mag * sinf(ph));
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ wsq,
                                                       float* __restrict__ wav, int T) {
    // four consecutive output samples per thread: hop 160, frame 640 and the 240-sample crop are multiples of 4, so the
    // four samples are covered by the same (up to 4) frames at 16-byte aligned offsets; sums in frame order as before
    const int b = blockIdx.y;
    const long n = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long L = (long)T * 160;
    if (n >= L) return;
    const long p = n + 240;
    long tlo = (p - 639 + 159) / 160;  // ceil((p-639)/160) for p-639 >= 0
    if (p - 639 < 0) tlo = 0;
    long thi = p / 160;
    if (thi > T - 1) thi = T - 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), env = acc;
    for (long t = tlo; t <= thi; ++t) {
        const int k = (int)(p - 160 * t);
        const float4 f = *reinterpret_cast<const float4*>(frames + ((long)b * T + t) * 640 + k);
        const float4 w = *reinterpret_cast<const float4*>(wsq + k);
        acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
        env.x += w.x; env.y += w.y; env.z += w.z; env.w += w.w;
    }
    *reinterpret_cast<float4*>(wav + (long)b * L + n) =
        make_float4(__fdiv_rn(acc.x, env.x), __fdiv_rn(acc.y, env.y), __fdiv_rn(acc.z, env.z), __fdiv_rn(acc.w, env.w));
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f32_to_bf16(x[i]);
}

template <typename InT>
__global__ void cast_fp8_kernel(const InT* __restrict__ x, unsigned* __restrict__ y, long n4, float scale, unsigned* sat) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 elements
    if (i >= n4) return;
    float a, b, c, d;
    if constexpr (sizeof(InT) == 4) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        a = v.x; b = v.y; c = v.z; d = v.w;
    } else {
        const uint2 v = reinterpret_cast<const uint2*>(x)[i];
        a = bf16_to_f32((bf16_t)(v.x & 0xffff)); b = bf16_to_f32((bf16_t)(v.x >> 16));
        c = bf16_to_f32((bf16_t)(v.y & 0xffff)); d = bf16_to_f32((bf16_t)(v.y >> 16));
    }
    float amax = 0.f;
    y[i] = fp8_pack4(a * scale, b * scale, c * scale, d * scale, amax);
    sat_commit(sat, 1, amax, SWC_FP8_LIMIT);
}

__global__ void cast_f16s_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ y, long rows,
                                 int K, float scale, unsigned* sat) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 logical columns
    const int k4 = K >> 2;
    if (i >= rows * k4) return;
    const long r = i / k4;
    const int k = (int)(i - r * k4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + k);
    float amax = 0.f;
    f16s_store4(y + r * 2L * K, k, v.x * scale, v.y * scale, v.z * scale, v.w * scale, amax);
    sat_commit(sat, 0, amax, SWC_F16S_LIMIT);
}

inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

#define OUT_DISPATCH(dt, CALL_F32, CALL_BF16) \
    do {                                      \
        if ((dt) == SWC_BF16) { CALL_BF16; } else { CALL_F32; } \
    } while (0)
#define OUT_DISPATCH3(dt, CALL_F32, CALL_BF16, CALL_F16S)                                        \
    do {                                                                                         \
        if ((dt) == SWC_BF16) { CALL_BF16; } else if ((dt) == SWC_F16S) { CALL_F16S; } else { CALL_F32; } \
    } while (0)

extern "C" int swc_layernorm(const float* x, void* y, const float* w, const float* b, const int32_t* lens,
                             int32_t B, int32_t t_in, int32_t t_out, int32_t C, float eps, int32_t y_dtype,
                             const int32_t* row_start, void* stream) {
    SWC_CHECK_ARG(x && y && w && b, "swc_layernorm: null pointer");
    SWC_CHECK_ARG(!row_start || lens, "swc_layernorm: packed input (row_start) needs lens");
    SWC_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * LN_MAXV, "swc_layernorm: C=%d unsupported", C);
    SWC_CHECK_ARG(y_dtype == SWC_F32 || y_dtype == SWC_BF16 || y_dtype == SWC_F16S || y_dtype == SWC_FP8,
                  "swc_layernorm: bad dtype");
    SWC_CHECK_ARG(y_dtype != SWC_F16S || C % 32 == 0, "swc_layernorm: split-f16 output needs C % 32 == 0");
    const int tw = t_out;  // every output row is written: rows beyond t_in are zeros
    const long rows = (long)B * tw;
    if (rows <= 0) return SWC_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((C == 512 || C == 768) && y_dtype != SWC_FP8) {
#define LN2_GO(OutT, NV)                                                                                             \
    hipLaunchKernelGGL((layernorm2_kernel<OutT, NV, 2>), dim3(nblk(rows, 8)), dim3(256), 0, s, x, (OutT*)y, w, b, lens, rows, \
                       tw, t_in, t_out, eps, swc_sat_counter(), row_start)  /* 4 rows per wave measured slower: 42 vs 47 % */
        if (C == 512) {
            OUT_DISPATCH3(y_dtype, LN2_GO(float, 2), LN2_GO(bf16_t, 2), LN2_GO(f16s_t, 2));
        } else {
            OUT_DISPATCH3(y_dtype, LN2_GO(float, 3), LN2_GO(bf16_t, 3), LN2_GO(f16s_t, 3));
        }
#undef LN2_GO
        SWC_CHECK_LAUNCH("swc_layernorm");
        return SWC_OK;
    }
    if (y_dtype == SWC_FP8) {
        hipLaunchKernelGGL(layernorm_kernel<fp8_t>, dim3(nblk(rows, 4)), dim3(256), 0, s, x, (fp8_t*)y, w, b, lens, rows,
                           tw, t_in, t_out, C, eps, swc_sat_counter(), row_start);
        SWC_CHECK_LAUNCH("swc_layernorm");
        return SWC_OK;
    }
    OUT_DISPATCH3(y_dtype,
                  hipLaunchKernelGGL(layernorm_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, s, x, (float*)y, w,
                                     b, lens, rows, tw, t_in, t_out, C, eps, swc_sat_counter(), row_start),
                  hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, s, x, (bf16_t*)y,
                                     w, b, lens, rows, tw, t_in, t_out, C, eps, swc_sat_counter(), row_start),
                  hipLaunchKernelGGL(layernorm_kernel<f16s_t>, dim3(nblk(rows, 4)), dim3(256), 0, s, x, (f16s_t*)y,
                                     w, b, lens, rows, tw, t_in, t_out, C, eps, swc_sat_counter(), row_start));
    SWC_CHECK_LAUNCH("swc_layernorm");
    return SWC_OK;
}

template <typename OutT, int NK, int S, bool FULL>
static int launch_dwconv7_ln(const float* x, void* y, const float* w, const float* bias, const float* ln_w,
                             const float* ln_b, int B, int T, int C, float eps, hipStream_t s) {
    const int lds = (S + 6) * C * 4;
    auto kern = dwconv7_ln_kernel<OutT, NK, S, FULL>;
    if (lds > 48 * 1024) SWC_ENABLE_LDS(kern, 160 * 1024, "swc_dwconv7_ln");
    const int nst = (T + S - 1) / S, nstrips = nst * B;
    // resident workgroups: LDS-limited per CU, 256 CUs; each walks nstrips / grid strips
    const int wgs = 0;
    const int fit = 160 * 1024 / lds;
    const int slots = 256 * (wgs > 0 ? wgs : (fit > 2 ? 2 : fit));  // <= 2 waves per SIMD by registers
    const int per = (nstrips + slots - 1) / slots;           // consecutive strips per workgroup
    const int grid = ((nstrips + per - 1) / per + 7) & ~7;   // multiple of 8: one contiguous strip range per XCD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, x, (OutT*)y, w, bias, ln_w, ln_b, T, C, nst, nstrips,
                       per, eps);
    return SWC_OK;
}

extern "C" int swc_dwconv7_ln(const float* x, void* y, const float* w, const float* bias, const float* ln_w,
                              const float* ln_b, int32_t B, int32_t T, int32_t C, float eps, int32_t y_dtype,
                              void* stream) {
    SWC_CHECK_ARG(x && y && w && bias && ln_w && ln_b, "swc_dwconv7_ln: null pointer");
    SWC_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * DW_MAXV, "swc_dwconv7_ln: C=%d unsupported", C);
    SWC_CHECK_ARG(y_dtype == SWC_F32 || y_dtype == SWC_BF16, "swc_dwconv7_ln: bad dtype");
    if ((long)B * T <= 0) return SWC_OK;
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define DW_GO(NK_, S_)                                                                                            \
    do {                                                                                                          \
        if (C == 256 * NK_)                                                                                       \
            rc = y_dtype == SWC_BF16                                                                              \
                     ? launch_dwconv7_ln<bf16_t, NK_, S_, true>(x, y, w, bias, ln_w, ln_b, B, T, C, eps, s)       \
                     : launch_dwconv7_ln<float, NK_, S_, true>(x, y, w, bias, ln_w, ln_b, B, T, C, eps, s);       \
        else                                                                                                      \
            rc = y_dtype == SWC_BF16                                                                              \
                     ? launch_dwconv7_ln<bf16_t, NK_, S_, false>(x, y, w, bias, ln_w, ln_b, B, T, C, eps, s)      \
                     : launch_dwconv7_ln<float, NK_, S_, false>(x, y, w, bias, ln_w, ln_b, B, T, C, eps, s);      \
    } while (0)
    // strips of 16 frames: (16+6) rows of LDS per workgroup, two workgroups per CU (register-limited)
    if (C <= 256) { DW_GO(1, 32); }
    else if (C <= 512) { DW_GO(2, 16); }
    else { DW_GO(4, 16); }
#undef DW_GO
    if (rc != SWC_OK) return rc;
    SWC_CHECK_LAUNCH("swc_dwconv7_ln");
    return SWC_OK;
}

extern "C" int swc_snake_aa(const float* x, void* y, const float* alpha, const float* beta,
                            const float* filt_host12, int32_t B, int32_t T, int32_t C, int32_t y_dtype,
                            void* stream) {
    SWC_CHECK_ARG(x && y && alpha && beta && filt_host12, "swc_snake_aa: null pointer");
    SWC_CHECK_ARG(y_dtype == SWC_F32 || y_dtype == SWC_BF16 || y_dtype == SWC_F16S, "swc_snake_aa: bad dtype");
    SWC_CHECK_ARG(y_dtype != SWC_F16S || C % 32 == 0, "swc_snake_aa: split-f16 output needs C % 32 == 0");
    SWC_CHECK_ARG(B <= 65535, "swc_snake_aa: B too large");
    if (B <= 0 || T <= 0 || C <= 0) return SWC_OK;
    Filt12 F;
    for (int i = 0; i < 12; ++i) F.f[i] = filt_host12[i];
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk(C, 256), nblk(T, 8), B);
    OUT_DISPATCH3(y_dtype,
                  hipLaunchKernelGGL((snake_aa_kernel<float, 8>), grid, dim3(256), 0, s, x, (float*)y, alpha, beta, F, T, C, swc_sat_counter()),
                  hipLaunchKernelGGL((snake_aa_kernel<bf16_t, 8>), grid, dim3(256), 0, s, x, (bf16_t*)y, alpha, beta, F, T, C, swc_sat_counter()),
                  hipLaunchKernelGGL((snake_aa_kernel<f16s_t, 8>), grid, dim3(256), 0, s, x, (f16s_t*)y, alpha, beta, F, T, C, swc_sat_counter()));
    SWC_CHECK_LAUNCH("swc_snake_aa");
    return SWC_OK;
}

static bool fsq_levels_ok(const int32_t* lv) {
    long prod = 1;
    for (int i = 0; i < 4; ++i) {
        if (lv[i] < 2 || lv[i] > 1024) return false;
        prod *= lv[i];
    }
    return prod < (1L << 31);   // the index of a group is an int32 code
}

extern "C" int swc_fsq_encode_levels(const float* z, int64_t ldz, float* zq, int32_t* codes, const int32_t* lens,
                                     const float* consts_host12, const int32_t* levels_host4, int32_t B, int32_t T,
                                     int32_t t_pad, int32_t G, void* stream) {
    SWC_CHECK_ARG(z && zq && codes && lens && consts_host12 && levels_host4, "swc_fsq_encode: null pointer");
    SWC_CHECK_ARG(G > 0 && ldz >= 4L * G && ldz % 4 == 0 && t_pad >= T, "swc_fsq_encode: bad shape");
    SWC_CHECK_ARG(aligned16(z) && aligned16(zq), "swc_fsq_encode: unaligned");
    SWC_CHECK_ARG(fsq_levels_ok(levels_host4), "swc_fsq_encode: levels must be 4 values in 2..1024 whose product fits an int32 code");
    const long total = (long)B * t_pad * G;
    if (total <= 0) return SWC_OK;
    FsqC K;
    for (int i = 0; i < 4; ++i) {
        K.scale[i] = consts_host12[i];
        K.offset[i] = consts_host12[4 + i];
        K.shift[i] = consts_host12[8 + i];
        K.levels[i] = levels_host4[i];
    }
    hipLaunchKernelGGL(fsq_encode_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, z, (long)ldz,
                       zq, codes, lens, B, T, t_pad, G, K);
    SWC_CHECK_LAUNCH("swc_fsq_encode");
    return SWC_OK;
}

extern "C" int swc_fsq_decode_levels(const int64_t* codes, float* zq, int64_t ldq, const int32_t* lens,
                                     const int32_t* levels_host4, int32_t B, int32_t T, int32_t G, void* stream) {
    SWC_CHECK_ARG(codes && zq && lens && levels_host4, "swc_fsq_decode: null pointer");
    SWC_CHECK_ARG(G > 0 && ldq >= 4L * G && ldq % 4 == 0, "swc_fsq_decode: bad shape");
    SWC_CHECK_ARG(aligned16(zq), "swc_fsq_decode: unaligned");
    SWC_CHECK_ARG(fsq_levels_ok(levels_host4), "swc_fsq_decode: levels must be 4 values in 2..1024 whose product fits an int32 code");
    const long total = (long)B * T * (ldq / 4);
    if (total <= 0) return SWC_OK;
    FsqC K = {};
    for (int i = 0; i < 4; ++i) K.levels[i] = levels_host4[i];
    hipLaunchKernelGGL(fsq_decode_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)codes, zq, (long)ldq, lens, B, T, G, K);
    SWC_CHECK_LAUNCH("swc_fsq_decode");
    return SWC_OK;
}

static const int32_t kShippedLevels[4] = {8, 7, 6, 6};   // config/SimWhisperCodec.yaml

extern "C" int swc_fsq_encode(const float* z, int64_t ldz, float* zq, int32_t* codes, const int32_t* lens,
                              const float* consts_host12, int32_t B, int32_t T, int32_t t_pad, int32_t G,
                              void* stream) {
    return swc_fsq_encode_levels(z, ldz, zq, codes, lens, consts_host12, kShippedLevels, B, T, t_pad, G, stream);
}

extern "C" int swc_fsq_decode(const int64_t* codes, float* zq, int64_t ldq, const int32_t* lens, int32_t B,
                              int32_t T, int32_t G, void* stream) {
    return swc_fsq_decode_levels(codes, zq, ldq, lens, kShippedLevels, B, T, G, stream);
}

extern "C" int swc_mel_frames(const float* wav, int64_t ld_wav, const int32_t* n, int32_t n_pad, float* frames,
                              int32_t B, int32_t T, void* stream) {
    SWC_CHECK_ARG(wav && n && frames, "swc_mel_frames: null pointer");
    SWC_CHECK_ARG(n_pad >= 400 && B <= 65535, "swc_mel_frames: bad n_pad/B");
    SWC_CHECK_ARG((long)(T - 1) * 160 + 199 < 2L * n_pad - 1, "swc_mel_frames: T too large for n_pad");
    if (B <= 0 || T <= 0) return SWC_OK;
    SWC_CHECK_ARG(aligned16(frames), "swc_mel_frames: frames not 16-byte aligned");
    const int vec_ok = aligned16(wav) && ld_wav % 4 == 0;  // rows start on 16-byte boundaries: interior quads are one load
    hipLaunchKernelGGL(mel_frames_kernel, dim3(nblk((long)T * 100, 256), B), dim3(256), 0, (hipStream_t)stream, wav,
                       (long)ld_wav, n, n_pad, frames, T, vec_ok);
    SWC_CHECK_LAUNCH("swc_mel_frames");
    return SWC_OK;
}

extern "C" int swc_mel_power(const float* dft, int64_t ld, float* pw, int64_t ldp, int64_t rows, void* stream) {
    SWC_CHECK_ARG(dft && pw && ld >= 402 && ldp >= 201, "swc_mel_power: bad args");
    if (rows <= 0) return SWC_OK;
    hipLaunchKernelGGL(mel_power_kernel, dim3(nblk(rows * ldp, 256)), dim3(256), 0, (hipStream_t)stream, dft,
                       (long)ld, pw, (long)ldp, (long)rows);
    SWC_CHECK_LAUNCH("swc_mel_power");
    return SWC_OK;
}

// umax is float on the API; internally compared through an order-preserving int map.
extern "C" int swc_mel_logmax(float* mel, int64_t ld, float* umax, int32_t B, int32_t T, int32_t n_mel,
                              void* stream) {
    SWC_CHECK_ARG(mel && umax && ld >= n_mel && B <= 65535, "swc_mel_logmax: bad args");
    if (B <= 0 || T <= 0) return SWC_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ordered_init_kernel, dim3(nblk(B, 256)), dim3(256), 0, s, umax, (int*)umax, B);
    hipLaunchKernelGGL(mel_logmax_kernel, dim3(nblk((long)T * n_mel, 256 * LM_PER_THREAD), B), dim3(256), 0, s, mel, (long)ld,
                       (int*)umax, T, n_mel);
    SWC_CHECK_LAUNCH("swc_mel_logmax");
    return SWC_OK;
}

extern "C" int swc_mel_final(const float* mel, int64_t ld, const float* umax, void* out, int64_t ldo, int32_t B,
                             int32_t T, int32_t n_mel, int32_t out_dtype, void* stream) {
    SWC_CHECK_ARG(mel && umax && out && ldo >= n_mel && B <= 65535, "swc_mel_final: bad args");
    if (B <= 0 || T <= 0) return SWC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk((long)T * ldo, 256), B);
    OUT_DISPATCH(out_dtype,
                 hipLaunchKernelGGL(mel_final_kernel<float>, grid, dim3(256), 0, s, mel, (long)ld, (const int*)umax,
                                    (float*)out, (long)ldo, T, n_mel),
                 hipLaunchKernelGGL(mel_final_kernel<bf16_t>, grid, dim3(256), 0, s, mel, (long)ld,
                                    (const int*)umax, (bf16_t*)out, (long)ldo, T, n_mel));
    SWC_CHECK_LAUNCH("swc_mel_final");
    return SWC_OK;
}

extern "C" int swc_deconv_col2im(const float* y3, const float* bias, void* out, int64_t ldo, int32_t B, int32_t T,
                                 int32_t C, int32_t s_, int32_t t_out, int32_t out_dtype, void* stream) {
    SWC_CHECK_ARG(y3 && bias && out && ldo >= C && s_ >= 1 && B <= 65535, "swc_deconv_col2im: bad args");
    SWC_CHECK_ARG(t_out <= (T - 1) * s_ + 3, "swc_deconv_col2im: t_out too large");
    if (B <= 0 || T <= 0 || t_out <= 0) return SWC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0 && ldo % 4 == 0 && aligned16(y3) && aligned16(bias) && aligned16(out)) {
        dim3 grid4(nblk((long)t_out * (ldo / 4), 256), B);
        OUT_DISPATCH(out_dtype,
                     hipLaunchKernelGGL(deconv_col2im4_kernel<float>, grid4, dim3(256), 0, s, y3, bias, (float*)out,
                                        (long)ldo, T, C, s_, t_out),
                     hipLaunchKernelGGL(deconv_col2im4_kernel<bf16_t>, grid4, dim3(256), 0, s, y3, bias, (bf16_t*)out,
                                        (long)ldo, T, C, s_, t_out));
        SWC_CHECK_LAUNCH("swc_deconv_col2im");
        return SWC_OK;
    }
    dim3 grid(nblk((long)t_out * ldo, 256), B);
    OUT_DISPATCH(out_dtype,
                 hipLaunchKernelGGL(deconv_col2im_kernel<float>, grid, dim3(256), 0, s, y3, bias, (float*)out,
                                    (long)ldo, T, C, s_, t_out),
                 hipLaunchKernelGGL(deconv_col2im_kernel<bf16_t>, grid, dim3(256), 0, s, y3, bias, (bf16_t*)out,
                                    (long)ldo, T, C, s_, t_out));
    SWC_CHECK_LAUNCH("swc_deconv_col2im");
    return SWC_OK;
}

extern "C" int swc_istft_spec(const float* h, int64_t ldh, void* sp, int64_t lds, int64_t rows, int32_t s_dtype,
                              void* stream) {
    SWC_CHECK_ARG(h && sp && ldh >= 642 && lds >= 642, "swc_istft_spec: bad args");
    SWC_CHECK_ARG(s_dtype != SWC_F16S || lds % 32 == 0, "swc_istft_spec: split-f16 rows need lds %% 32 == 0 (lds=%ld)", (long)lds);
    if (rows <= 0) return SWC_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk(rows * (321 + (lds - 642)), 256));
    OUT_DISPATCH3(s_dtype,
                  hipLaunchKernelGGL(istft_spec2_kernel<float>, grid, dim3(256), 0, s, h, (long)ldh, (float*)sp,
                                     (long)lds, (long)rows),
                  hipLaunchKernelGGL(istft_spec2_kernel<bf16_t>, grid, dim3(256), 0, s, h, (long)ldh, (bf16_t*)sp,
                                     (long)lds, (long)rows),
                  hipLaunchKernelGGL(istft_spec2_kernel<f16s_t>, grid, dim3(256), 0, s, h, (long)ldh, (f16s_t*)sp,
                                     (long)lds, (long)rows));
    SWC_CHECK_LAUNCH("swc_istft_spec");
    return SWC_OK;
}

extern "C" int swc_istft_ola(const float* frames, const float* window_sq, float* wav, int32_t B, int32_t T,
                             void* stream) {
    SWC_CHECK_ARG(frames && window_sq && wav && B <= 65535, "swc_istft_ola: bad args");
    if (B <= 0 || T <= 0) return SWC_OK;
    SWC_CHECK_ARG(aligned16(frames) && aligned16(window_sq) && aligned16(wav), "swc_istft_ola: unaligned");
    hipLaunchKernelGGL(istft_ola_kernel, dim3(nblk((long)T * 40, 256), B), dim3(256), 0, (hipStream_t)stream,
                       frames, window_sq, wav, T);
    SWC_CHECK_LAUNCH("swc_istft_ola");
    return SWC_OK;
}

extern "C" int swc_cast_f32_f16s(const float* x, int64_t ldx, void* y, int64_t rows, int32_t K, float scale,
                                 void* stream) {
    SWC_CHECK_ARG(x && y, "swc_cast_f32_f16s: null pointer");
    SWC_CHECK_ARG(K > 0 && K % 32 == 0 && ldx >= K && ldx % 4 == 0 && aligned16(x) && aligned16(y),
                  "swc_cast_f32_f16s: K=%d must be a multiple of 32, rows 16-byte aligned", K);
    if (rows <= 0) return SWC_OK;
    hipLaunchKernelGGL(cast_f16s_kernel, dim3(nblk(rows * (K / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                       (unsigned short*)y, (long)rows, K, scale == 0.0f ? 1.0f : scale, swc_sat_counter());
    SWC_CHECK_LAUNCH("swc_cast_f32_f16s");
    return SWC_OK;
}

extern "C" int swc_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
    SWC_CHECK_ARG(x && y, "swc_cast_f32_bf16: null pointer");
    if (n <= 0) return SWC_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y,
                       (long)n);
    SWC_CHECK_LAUNCH("swc_cast_f32_bf16");
    return SWC_OK;
}

namespace {
// grid (chunks, rows): dword copy of row r with zero fill up to ld (16-byte vectors when everything is aligned)
__global__ void gather_rows_kernel(const void* const* __restrict__ src, const int64_t* __restrict__ nbytes,
                                   unsigned* __restrict__ out, long ld_words) {
    const int r = blockIdx.y;
    const unsigned* s = reinterpret_cast<const unsigned*>(src[r]);
    const long n = nbytes[r] >> 2;
    unsigned* o = out + (long)r * ld_words;
    const bool v4 = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(o)) & 15) == 0 && (ld_words & 3) == 0;
    const long stride = (long)gridDim.x * blockDim.x;
    if (v4) {
        const long n4 = n >> 2, l4 = ld_words >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < l4; i += stride) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < n4) {
                v = reinterpret_cast<const uint4*>(s)[i];
            } else if (4 * i < n) {  // the row ends inside this vector
                unsigned t[4] = {0u, 0u, 0u, 0u};
                for (int k = 0; k < 4 && 4 * i + k < n; ++k) t[k] = s[4 * i + k];
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            reinterpret_cast<uint4*>(o)[i] = v;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ld_words; i += stride) o[i] = i < n ? s[i] : 0u;
    }
}
}  // namespace

extern "C" int swc_gather_rows(const void* const* src, const int64_t* nbytes, void* out, int64_t ld_bytes,
                               int32_t n_rows, void* stream) {
    SWC_CHECK_ARG(src && nbytes && out, "swc_gather_rows: null pointer");
    SWC_CHECK_ARG(ld_bytes >= 0 && ld_bytes % 4 == 0 && n_rows >= 0 && n_rows <= 65535, "swc_gather_rows: bad shape");
    if (n_rows == 0 || ld_bytes == 0) return SWC_OK;
    const long words = ld_bytes / 4;
    long chunks = nblk(nblk(words, 4), 256);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)chunks, n_rows), dim3(256), 0, (hipStream_t)stream, src,
                       nbytes, (unsigned*)out, words);
    SWC_CHECK_LAUNCH("swc_gather_rows");
    return SWC_OK;
}

extern "C" int swc_cast_fp8(const void* x, int32_t x_dtype, void* y, int64_t n, float scale, void* stream) {
    SWC_CHECK_ARG(x && y, "swc_cast_fp8: null pointer");
    SWC_CHECK_ARG(x_dtype == SWC_F32 || x_dtype == SWC_BF16, "swc_cast_fp8: bad x_dtype");
    SWC_CHECK_ARG(n >= 0 && n % 4 == 0, "swc_cast_fp8: n=%ld not a multiple of 4", (long)n);
    SWC_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 3) == 0,
                  "swc_cast_fp8: unaligned");
    if (n == 0) return SWC_OK;
    const long n4 = n / 4;
    if (x_dtype == SWC_F32)
        hipLaunchKernelGGL(cast_fp8_kernel<float>, dim3(nblk(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (unsigned*)y, n4, scale, swc_sat_counter());
    else
        hipLaunchKernelGGL(cast_fp8_kernel<bf16_t>, dim3(nblk(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (unsigned*)y, n4, scale, swc_sat_counter());
    SWC_CHECK_LAUNCH("swc_cast_fp8");
    return SWC_OK;
}

// ---------------------------------------------------------------- PCM16 <-> f32 (file IO around the path)
// Both directions are HBM-bound streams (6 bytes per sample); 8 samples per thread, 16-byte accesses on the wide side.
namespace {
__global__ void pcm16_to_f32_kernel(const short* __restrict__ in, float* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n && ((reinterpret_cast<uintptr_t>(in + i) | reinterpret_cast<uintptr_t>(out + i)) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = (float)(short)(w[k] & 0xFFFFu) * 0x1p-15f;
            f[2 * k + 1] = (float)(short)(w[k] >> 16) * 0x1p-15f;
        }
        reinterpret_cast<float4*>(out + i)[0] = make_float4(f[0], f[1], f[2], f[3]);
        reinterpret_cast<float4*>(out + i)[1] = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        for (long k = i; k < n && k < i + 8; ++k) out[k] = (float)in[k] * 0x1p-15f;
    }
}
__device__ __forceinline__ int pcm16_of(float x) {
    // round(clip(x, -1, 1) * 32767), ties to even, every step in f32: numpy's result on the host (wavio.save_audio);
    // a NaN sample (the path never produces one) becomes -32767 here
    return __float2int_rn(__fmul_rn(fminf(fmaxf(x, -1.0f), 1.0f), 32767.0f));
}
__global__ void f32_to_pcm16_kernel(const float* __restrict__ in, short* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n && ((reinterpret_cast<uintptr_t>(in + i) | reinterpret_cast<uintptr_t>(out + i)) & 15) == 0) {
        const float4 a = reinterpret_cast<const float4*>(in + i)[0], b = reinterpret_cast<const float4*>(in + i)[1];
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = ((unsigned)pcm16_of(f[2 * k]) & 0xFFFFu) | ((unsigned)pcm16_of(f[2 * k + 1]) << 16);
        *reinterpret_cast<uint4*>(out + i) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (long k = i; k < n && k < i + 8; ++k) out[k] = (short)pcm16_of(in[k]);
    }
}
}  // namespace

extern "C" int swc_pcm16_to_f32(const int16_t* pcm, float* out, int64_t n, void* stream) {
    SWC_CHECK_ARG(n >= 0 && (n == 0 || (pcm && out)), "swc_pcm16_to_f32: bad args");
    if (n == 0) return SWC_OK;
    hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3(nblk(nblk(n, 8), 256)), dim3(256), 0, (hipStream_t)stream, (const short*)pcm,
                       out, (long)n);
    SWC_CHECK_LAUNCH("swc_pcm16_to_f32");
    return SWC_OK;
}

extern "C" int swc_f32_to_pcm16(const float* x, int16_t* pcm, int64_t n, void* stream) {
    SWC_CHECK_ARG(n >= 0 && (n == 0 || (x && pcm)), "swc_f32_to_pcm16: bad args");
    if (n == 0) return SWC_OK;
    hipLaunchKernelGGL(f32_to_pcm16_kernel, dim3(nblk(nblk(n, 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (short*)pcm,
                       (long)n);
    SWC_CHECK_LAUNCH("swc_f32_to_pcm16");
    return SWC_OK;
}

// ---------------------------------------------------------------- valid-token packing
// [B][T][row_bytes] padded rows -> packed rows: utterance b's first lens[b] rows land at row_start[b] (16-byte chunks).
namespace {
__global__ void pack_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int* __restrict__ row_start,
                                 const int* __restrict__ lens, int T, int chunks) {
    const int b = blockIdx.y;
    const int n = lens[b] < T ? lens[b] : T;
    const long total = (long)n * chunks;
    const uint4* s = src + (long)b * T * chunks;
    uint4* d = dst + (long)row_start[b] * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}
}  // namespace

extern "C" int swc_pack_rows(const void* src, void* dst, const int32_t* row_start, const int32_t* lens, int32_t B, int32_t T,
                             int64_t row_bytes, void* stream) {
    SWC_CHECK_ARG(src && dst && row_start && lens, "swc_pack_rows: null pointer");
    SWC_CHECK_ARG(B >= 0 && B <= 65535 && T >= 0 && row_bytes > 0 && row_bytes % 16 == 0 && aligned16(src) && aligned16(dst),
                  "swc_pack_rows: bad shape or alignment");
    if (B == 0 || T == 0) return SWC_OK;
    const long per = (long)T * (row_bytes / 16);
    const unsigned gx = (unsigned)((per + 255) / 256 > 1024 ? 1024 : (per + 255) / 256);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       row_start, lens, T, (int)(row_bytes / 16));
    SWC_CHECK_LAUNCH("swc_pack_rows");
    return SWC_OK;
}

// ---------------------------------------------------------------- phase shift between two streams
// One wave that sleeps for `us` microseconds of the 100 MHz real-time counter and exits (bounded by construction).  Used
// to start the second of two chains of equal-length launches half a launch late, so that the chains' memory phases fall
// into each other's compute phases (codec._blocks_two_streams).
namespace {
__global__ void delay_kernel(long ticks) {
    const long t0 = (long)__builtin_amdgcn_s_memrealtime();
    for (int guard = 0; guard < (1 << 22); ++guard) {  // at most ~4 M sleeps: the loop ends even with a stuck counter
        if ((long)__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}
}  // namespace

extern "C" int swc_delay_us(int32_t us, void* stream) {
    SWC_CHECK_ARG(us >= 0 && us <= 100000, "swc_delay_us: 0 <= us <= 100000");
    if (us == 0) return SWC_OK;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)us * 100);
    SWC_CHECK_LAUNCH("swc_delay_us");
    return SWC_OK;
}

// ---------------------------------------------------------------- code bitstream
// 8 groups x 11 bits (2016 < 2^11 codes per group) = 88 bits = 11 bytes per 12.5 Hz frame: 1100 bit/s, the
// codec's nominal bitrate.  Frame t occupies bytes [11 t, 11 t + 11); group g occupies bits [11 g, 11 g + 11) of
// the frame, least-significant bit first.  The reference keeps codes in memory only (model.py:302).
namespace {
__global__ void codes_pack_kernel(const int* __restrict__ codes, long ldg, unsigned char* __restrict__ out, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    unsigned long long lo = 0;  // bits 0..63
    unsigned int hi = 0;        // bits 64..87
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const unsigned long long c = (unsigned)codes[g * ldg + t] & 0x7ffu;
        const int sh = 11 * g;
        if (sh < 64) lo |= c << sh;
        if (sh + 11 > 64) hi |= (unsigned)(sh >= 64 ? c << (sh - 64) : c >> (64 - sh));
    }
    unsigned char* p = out + 11L * t;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (unsigned char)(lo >> (8 * i));
#pragma unroll
    for (int i = 0; i < 3; ++i) p[8 + i] = (unsigned char)(hi >> (8 * i));
}
__global__ void codes_unpack_kernel(const unsigned char* __restrict__ in, int* __restrict__ codes, long ldg, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const unsigned char* p = in + 11L * t;
    unsigned long long lo = 0;
    unsigned int hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) lo |= (unsigned long long)p[i] << (8 * i);
#pragma unroll
    for (int i = 0; i < 3; ++i) hi |= (unsigned)p[8 + i] << (8 * i);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int sh = 11 * g;
        unsigned long long v = sh < 64 ? lo >> sh : 0;
        if (sh + 11 > 64) v |= sh >= 64 ? (unsigned long long)(hi >> (sh - 64)) : (unsigned long long)hi << (64 - sh);
        codes[g * ldg + t] = (int)(v & 0x7ffu);
    }
}
}  // namespace

extern "C" int swc_codes_pack(const int32_t* codes, int64_t ldg, void* bytes, int32_t T, void* stream) {
    if (T == 0) return SWC_OK;
    SWC_CHECK_ARG(codes && bytes && ldg >= T && T > 0, "swc_codes_pack: bad args");
    hipLaunchKernelGGL(codes_pack_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, codes, (long)ldg,
                       (unsigned char*)bytes, T);
    SWC_CHECK_LAUNCH("swc_codes_pack");
    return SWC_OK;
}

extern "C" int swc_codes_unpack(const void* bytes, int32_t* codes, int64_t ldg, int32_t T, void* stream) {
    if (T == 0) return SWC_OK;
    SWC_CHECK_ARG(codes && bytes && ldg >= T && T > 0, "swc_codes_unpack: bad args");
    hipLaunchKernelGGL(codes_unpack_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)bytes, codes, (long)ldg, T);
    SWC_CHECK_LAUNCH("swc_codes_unpack");
    return SWC_OK;
}
