// Does VALU work overlap with MFMA work on a gfx950 SIMD — inside one wave, and between two waves of one SIMD?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap tools/probes/mfma_valu_overlap.hip ; run: /tmp/overlap
// Each kernel runs ITER iterations of a fixed instruction mix in inline asm (so that hipcc cannot reorder or drop it) and
// reports cycles per iteration (s_memtime of wave 0 of workgroup 0, 100 MHz counter scaled by wall clock elsewhere; here we
// simply use wall time over many workgroups and print ns per iteration and the implied cycles at 2.4 GHz).
//   mode 0: 16 MFMA (16x16x32 f16, 4 independent accumulators)                       -> matrix pipe alone
//   mode 1: 48 v_fma_f32 (independent)                                                -> VALU alone
//   mode 2: 16 x (1 MFMA + 3 v_fma_f32) interleaved in ONE wave                       -> in-wave overlap
//   mode 3: two waves per SIMD: waves 0-3 run mode 0, waves 4-7 run mode 1 (512 threads)  -> cross-wave overlap
//   mode 4: two waves per SIMD, both run mode 0 then mode 1 in the same order (the "same phase" case)
//   mode 5: 48 v_exp_f32   (transcendental rate)
//   mode 6: 16 x (1 MFMA + 1 v_exp_f32 + 2 v_fma) in one wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA16B(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
#define MFMA32B(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(0.5f); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x * 0.01f + i;
    float c1 = 0.999f, c2 = 0.001f;
    // pseudo-random bf16 operands (the clock the chip holds depends on the data: zeros run faster than random bits)
    bf16x8 ab, bb;
    {
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        unsigned short t[16];
        for (int i = 0; i < 16; ++i) { h = h * 1664525u + 1013904223u; t[i] = (unsigned short)(0x3c00u | ((h >> 9) & 0x83ffu)); }
        for (int i = 0; i < 8; ++i) { ab[i] = __builtin_bit_cast(__bf16, t[i]); bb[i] = __builtin_bit_cast(__bf16, t[8 + i]); }
    }
    f32x16 big0 = {0}, big1 = big0, big2 = big0, big3 = big0;
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 3 && !second)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        } else if (MODE == 1 || (MODE == 3 && second)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) FMA(x[i]);
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                MFMA(acc0); FMA(x[0]); FMA(x[1]); FMA(x[2]);
                MFMA(acc1); FMA(x[3]); FMA(x[4]); FMA(x[5]);
                MFMA(acc2); FMA(x[6]); FMA(x[7]); FMA(x[8]);
                MFMA(acc3); FMA(x[9]); FMA(x[10]); FMA(x[11]);
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) FMA(x[i]);
        } else if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                MFMA(acc0); FMA(x[0]); FMA(x[1]);
                MFMA(acc1); FMA(x[3]); FMA(x[4]);
                MFMA(acc2); FMA(x[6]); FMA(x[7]);
                MFMA(acc3); FMA(x[9]); FMA(x[10]);
            }
        } else if (MODE == 8) {   // two waves per SIMD: the MFMA wave at raised priority
            if (!second) {
                __builtin_amdgcn_s_setprio(3);
#pragma unroll
                for (int r = 0; r < 4; ++r) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 12; ++i) FMA(x[i]);
            }
        } else if (MODE == 9) {   // two waves per SIMD: both run the in-wave interleave
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                MFMA(acc0); FMA(x[0]); FMA(x[1]); FMA(x[2]);
                MFMA(acc1); FMA(x[3]); FMA(x[4]); FMA(x[5]);
                MFMA(acc2); FMA(x[6]); FMA(x[7]); FMA(x[8]);
                MFMA(acc3); FMA(x[9]); FMA(x[10]); FMA(x[11]);
            }
        } else if (MODE == 10) {  // ONE accumulation chain
#pragma unroll
            for (int r = 0; r < 16; ++r) MFMA(acc0);
        } else if (MODE == 11) {  // two chains
#pragma unroll
            for (int r = 0; r < 8; ++r) { MFMA(acc0); MFMA(acc1); }
        } else if (MODE == 12) {  // three chains (15 MFMAs + 1)
#pragma unroll
            for (int r = 0; r < 5; ++r) { MFMA(acc0); MFMA(acc1); MFMA(acc2); }
            MFMA(acc3);
        } else if (MODE == 13) {  // 16 x 16x16x32 bf16, random operands, 4 accumulators
#pragma unroll
            for (int r = 0; r < 4; ++r) { MFMA16B(acc0); MFMA16B(acc1); MFMA16B(acc2); MFMA16B(acc3); }
        } else if (MODE == 14) {  // 8 x 32x32x16 bf16 (the same flops), random operands, 4 accumulators
#pragma unroll
            for (int r = 0; r < 2; ++r) { MFMA32B(big0); MFMA32B(big1); MFMA32B(big2); MFMA32B(big3); }
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) EXP(x[i]);
        } else if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                MFMA(acc0); EXP(x[0]); FMA(x[1]); FMA(x[2]);
                MFMA(acc1); EXP(x[3]); FMA(x[4]); FMA(x[5]);
                MFMA(acc2); EXP(x[6]); FMA(x[7]); FMA(x[8]);
                MFMA(acc3); EXP(x[9]); FMA(x[10]); FMA(x[11]);
            }
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 12; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    for (int i = 0; i < 16; ++i) s += big0[i] + big1[i] + big2[i] + big3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what, int threads, float* d, unsigned long long* cyc) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    unsigned long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d  %-62s %8.1f ns/iter | shader cycles/iter: wave 0 %7.1f", MODE, what, ns, (double)h[0] / iters);
    if (threads == 512) printf("  wave 4 %7.1f", (double)h[4] / iters);
    printf("   (%.2f GHz)\n", (double)h[0] / iters / ns);
}

int main() {
    float* d;
    unsigned long long* cyc;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 8 * sizeof(unsigned long long));
    run<0>("16 MFMA, one wave per SIMD", 256, d, cyc);
    run<1>("48 v_fma, one wave per SIMD", 256, d, cyc);
    run<2>("16 x (MFMA + 3 v_fma) in one wave", 256, d, cyc);
    run<3>("two waves per SIMD: one 16 MFMA, the other 48 v_fma", 512, d, cyc);
    run<4>("two waves per SIMD, both: 16 MFMA then 48 v_fma", 512, d, cyc);
    run<4>("one wave per SIMD: 16 MFMA then 48 v_fma", 256, d, cyc);
    run<5>("48 v_exp, one wave per SIMD", 256, d, cyc);
    run<6>("16 x (MFMA + v_exp + 2 v_fma) in one wave", 256, d, cyc);
    run<0>("16 MFMA, two waves per SIMD (each)", 512, d, cyc);
    run<1>("48 v_fma, two waves per SIMD (each)", 512, d, cyc);
    run<10>("16 MFMA on ONE accumulator (dependent chain)", 256, d, cyc);
    run<11>("16 MFMA on two accumulators", 256, d, cyc);
    run<12>("16 MFMA on three accumulators", 256, d, cyc);
    run<10>("16 MFMA on ONE accumulator, two waves per SIMD", 512, d, cyc);
    run<11>("16 MFMA on two accumulators, two waves per SIMD", 512, d, cyc);
    run<13>("16 MFMA 16x16x32 bf16, random operands", 256, d, cyc);
    run<14>("8 MFMA 32x32x16 bf16 (same flops), random operands", 256, d, cyc);
    run<13>("16 MFMA 16x16x32 bf16, random, two waves per SIMD", 512, d, cyc);
    run<14>("8 MFMA 32x32x16 bf16, random, two waves per SIMD", 512, d, cyc);
    run<7>("16 x (MFMA + 2 v_fma) in one wave", 256, d, cyc);
    run<8>("two waves per SIMD: 16 MFMA at s_setprio 3 | 48 v_fma", 512, d, cyc);
    run<9>("two waves per SIMD, both 16 x (MFMA + 3 v_fma)", 512, d, cyc);
    return 0;
}
