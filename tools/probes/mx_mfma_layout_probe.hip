#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A: 16 rows x 128 k (fp8 e4m3), B: 128 k x 16 cols.  Each lane supplies 32 bytes of A and of B.
// Probe: lane la supplies byte ja of A = 1.0 (0x38), everything else 0; lane lb supplies byte jb of B = 1.0.
// D[row][col] != 0  <=>  (row, k) of A's element == (k, col) of B's element  -> learn the maps.
__global__ void probe(const unsigned char* A, const unsigned char* B, float* D, int scale_a, int scale_b) {
    int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + lane * 32))[i]; b[i] = ((const int*)(B + lane * 32))[i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; ++r) D[lane * 4 + r] = c[r];
}
int main() {
    unsigned char *A, *B; float* D;
    (void)hipHostMalloc(&A, 64 * 32); (void)hipHostMalloc(&B, 64 * 32); (void)hipHostMalloc(&D, 64 * 4 * 4);
    // encode k index in values: A[lane][j] = 1.0 only for one (lane, j) at a time is slow (2048^2); instead use distinct powers:
    // step 1: B all ones (1.0), A one-hot at (la, ja): D row = which row A's element belongs to (all cols non-zero).
    // step 2: A all ones, B one-hot at (lb, jb): D col.
    // step 3: k pairing: A one-hot (la, ja), B one-hot (lb, jb) with same k => need k map: use A one-hot, B[lane][j] = value v(k?) unknown...
    // Use: A one-hot at (la, ja) (value 1.0); B = all (lb, jb) set to distinct values is impossible in fp8; so do k pairing by
    // brute force over lb in 0..63, jb in 0..31 for a few (la, ja).
    int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
    printf("A row map (la, ja -> row), B col map:\n");
    for (int la = 0; la < 64; la += 1) {
        for (int ja = 0; ja < 32; ja += 8) {
            memset(A, 0, 2048); memset(B, 0x38, 2048); A[la * 32 + ja] = 0x38;
            probe<<<1, 64>>>(A, B, D, sa, sb); (void)hipDeviceSynchronize();
            int row = -1, cnt = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l * 4 + r] != 0) { row = (l >> 4) * 4 + r; cnt++; }
            if (ja == 0 && (la < 20 || la % 16 == 0)) printf(" A lane %2d byte %2d -> row %2d (%d outputs, val %g)\n", la, ja, row, cnt, D[0]);
        }
    }
    // k map: for A (la=0, ja) find which (lb, jb) of B pairs with it (B one-hot, col = lb&15 presumably)
    for (int la : {0, 16, 32, 48, 5}) for (int ja : {0, 1, 4, 8, 15, 16, 31}) {
        memset(A, 0, 2048); A[la * 32 + ja] = 0x38;
        int found = 0;
        for (int lb = 0; lb < 64 && !found; lb += 16) for (int jb = 0; jb < 32 && !found; ++jb) {
            memset(B, 0, 2048); B[lb * 32 + jb] = 0x38;
            probe<<<1, 64>>>(A, B, D, sa, sb); (void)hipDeviceSynchronize();
            for (int i = 0; i < 256; ++i) if (D[i] != 0) { printf(" A(lane %2d, byte %2d) pairs with B(lane %2d, byte %2d): D idx lane %d reg %d = %g\n", la, ja, lb, jb, i / 4, i % 4, D[i]); found = 1; break; }
        }
        if (!found) printf(" A(lane %d, byte %d): no partner among lanes 0,16,32,48\n", la, ja);
    }
    // scale semantics: scale_a = 0x80 in byte 0 (x2) for all lanes
    memset(A, 0x38, 2048); memset(B, 0x38, 2048);
    probe<<<1, 64>>>(A, B, D, 0x7f7f7f7f, 0x7f7f7f7f); (void)hipDeviceSynchronize(); printf("all ones: D[0] = %g (expect 128)\n", D[0]);
    probe<<<1, 64>>>(A, B, D, 0x7f7f7f80, 0x7f7f7f7f); (void)hipDeviceSynchronize(); printf("scale_a byte0 = 0x80: D[0] = %g\n", D[0]);
    probe<<<1, 64>>>(A, B, D, 0x807f7f7f, 0x7f7f7f7f); (void)hipDeviceSynchronize(); printf("scale_a byte3 = 0x80: D[0] = %g\n", D[0]);
    return 0;
}
