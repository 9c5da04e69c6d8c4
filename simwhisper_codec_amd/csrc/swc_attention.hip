// swc_attention: varlen, non-causal, head_dim 64 flash attention on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32).  One workgroup = 64 queries of one (utterance, head); each of
// its 4 waves owns 16 queries and walks the valid keys in tiles of 64.
//
// The score tile is computed TRANSPOSED (S^T = K Q^T): its C/D fragment (col = query on
// lane & 15, row = key on (lane >> 4) * 4 + reg) is then already the B operand of the
// second product O^T = V^T P^T, so P never leaves registers, the softmax statistics of a
// query live on one lane column (2 xor-shuffles for a row maximum), and the O^T fragment
// gives every lane 4 contiguous output channels (one 16-byte store).
#include "swc_common.h"

namespace {

constexpr int HD = 64;        // head dim
constexpr int KT = 64;        // keys per tile
constexpr int LDK = 68;       // padded LDS row (floats)
constexpr int QB = 64;        // queries per workgroup

template <bool BF16>
__device__ __forceinline__ float4 load4(const void* base, long idx) {
    if constexpr (BF16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + idx);
        float4 r;
        r.x = __uint_as_float(u.x << 16);
        r.y = __uint_as_float(u.x & 0xffff0000u);
        r.z = __uint_as_float(u.y << 16);
        r.w = __uint_as_float(u.y & 0xffff0000u);
        return r;
    } else {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
    }
}

template <bool BF16, bool OUT_F16S = false>
__global__ __launch_bounds__(256, 2) void attn_kernel(const void* __restrict__ qkv,
                                                      void* __restrict__ out,
                                                      const int* __restrict__ lens, int T, int H) {
    __shared__ __attribute__((aligned(16))) float sK[KT * LDK];
    __shared__ __attribute__((aligned(16))) float sV[KT * LDK];

    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * QB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fh = lane >> 4;
    const int D = H * HD;
    const long ld = 3L * D;
    int len = lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);

    const int q = q0 + wave * 16 + fr;  // this lane's query column
    const bool q_in = q < T;
    const long orow = ((long)b * T + (q_in ? q : T - 1)) * D + head * HD;
    // split-f16 output rows are addressed in halves (2 per logical column)
    unsigned short* orow_s = reinterpret_cast<unsigned short*>(out) + ((long)b * T + (q_in ? q : T - 1)) * 2L * D;

    if (q0 >= len) {  // whole block is padding: defined, finite output
        if (q_in) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if constexpr (OUT_F16S) {
                    f16s_store4(orow_s, head * HD + dt * 16 + fh * 4, 0.f, 0.f, 0.f, 0.f);
                } else if constexpr (BF16) {
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + orow + dt * 16 + fh * 4) =
                        make_uint2(0, 0);
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + orow + dt * 16 + fh * 4) =
                        make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        return;
    }

    // Q fragment: B operand of S^T, element (g, e) is d = 16g + 4*fh + e
    float qf[16];
    {
        const long qrow = ((long)b * T + (q_in ? q : T - 1)) * ld + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = load4<BF16>(qkv, qrow + 16 * g + 4 * fh);
            qf[4 * g + 0] = v.x; qf[4 * g + 1] = v.y; qf[4 * g + 2] = v.z; qf[4 * g + 3] = v.w;
        }
    }

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int ntile = (len + KT - 1) / KT;
    for (int kt = 0; kt < ntile; ++kt) {
        const int k0 = kt * KT;
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 4, c4 = c & 15;
            const int key = k0 + row;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (key < len) {
                const long base = ((long)b * T + key) * ld + head * HD + c4 * 4;
                kv = load4<BF16>(qkv, base + D);
                vv = load4<BF16>(qkv, base + 2 * D);
            }
            *reinterpret_cast<float4*>(&sK[row * LDK + c4 * 4]) = kv;
            *reinterpret_cast<float4*>(&sV[row * LDK + c4 * 4]) = vv;
        }
        __syncthreads();

        // S^T[key][q] for 4 key sub-tiles
        f32x4 s[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* kr = &sK[(ks * 16 + fr) * LDK + 4 * fh];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 kf = *reinterpret_cast<const float4*>(kr + 16 * g);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[4 * g + 0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[4 * g + 1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[4 * g + 2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[4 * g + 3], a, 0, 0, 0);
            }
            s[ks] = a;
        }
        // mask + tile max (keys of this lane: k0 + 16ks + 4fh + e)
        float mt = -INFINITY;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = k0 + ks * 16 + fh * 4 + e;
                float v = s[ks][e];
                v = key < len ? v : -INFINITY;
                s[ks][e] = v;
                mt = fmaxf(mt, v);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);   // finite: every tile holds >= 1 valid key
        const float alpha = expf(m_run - m_new);  // 0 on the first tile
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = expf(s[ks][e] - m_new);
                s[ks][e] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;  // per-lane partial (this lane's keys only)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;

        // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* vr = &sV[(ks * 16 + fh * 4 + e) * LDK + fr];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dt * 16], s[ks][e], o[dt], 0, 0, 0);
            }
    }

    float l = l_run;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (q_in) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = o[dt] * inv;
            if constexpr (OUT_F16S) {
                f16s_store4(orow_s, head * HD + dt * 16 + fh * 4, v[0] * SWC_F16S_ACT_SCALE, v[1] * SWC_F16S_ACT_SCALE,
                            v[2] * SWC_F16S_ACT_SCALE, v[3] * SWC_F16S_ACT_SCALE);
            } else if constexpr (BF16) {
                uint2 u;
                u.x = bf16_pack2(v[0], v[1]);
                u.y = bf16_pack2(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + orow + dt * 16 + fh * 4) = u;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + orow + dt * 16 + fh * 4) =
                    make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace

extern "C" int swc_attention(const void* qkv, void* out, const int32_t* lens, int32_t B, int32_t T,
                             int32_t H, int32_t dtype, void* stream) {
    SWC_CHECK_ARG(qkv && out && lens, "swc_attention: null pointer");
    SWC_CHECK_ARG(B >= 0 && T >= 0 && H > 0, "swc_attention: bad B/T/H");
    SWC_CHECK_ARG(dtype == SWC_F32 || dtype == SWC_BF16, "swc_attention: bad dtype");
    SWC_CHECK_ARG(aligned16(qkv) && aligned16(out), "swc_attention: unaligned");
    SWC_CHECK_ARG(B <= 65535 && H <= 65535, "swc_attention: B/H exceed grid limits");
    if (B == 0 || T == 0) return SWC_OK;
    dim3 grid((T + QB - 1) / QB, H, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWC_BF16)
        hipLaunchKernelGGL(attn_kernel<true>, grid, block, 0, s, qkv, out, lens, T, H);
    else
        hipLaunchKernelGGL(attn_kernel<false>, grid, block, 0, s, qkv, out, lens, T, H);
    SWC_CHECK_LAUNCH("swc_attention");
    return SWC_OK;
}

extern "C" int swc_attention_ex(const void* qkv, void* out, const int32_t* lens, int32_t B, int32_t T, int32_t H,
                                int32_t out_dtype, void* stream) {
    if (out_dtype == SWC_F32) return swc_attention(qkv, out, lens, B, T, H, SWC_F32, stream);
    SWC_CHECK_ARG(out_dtype == SWC_F16S, "swc_attention_ex: out_dtype must be F32 or F16S");
    SWC_CHECK_ARG(qkv && out && lens, "swc_attention_ex: null pointer");
    SWC_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && B <= 65535 && H <= 65535, "swc_attention_ex: bad B/T/H");
    SWC_CHECK_ARG(aligned16(qkv) && aligned16(out), "swc_attention_ex: unaligned");
    if (B == 0 || T == 0) return SWC_OK;
    dim3 grid((T + QB - 1) / QB, H, B), block(256);
    hipLaunchKernelGGL((attn_kernel<false, true>), grid, block, 0, (hipStream_t)stream, qkv, out, lens, T, H);
    SWC_CHECK_LAUNCH("swc_attention_ex");
    return SWC_OK;
}
