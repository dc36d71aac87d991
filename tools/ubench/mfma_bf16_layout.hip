// A = I-like check of the A/B operand layout of v_mfma_f32_32x32x16_bf16 on gfx950 (the evaluation's bf16 filter, DESIGN s8):
// hypothesis  A: lane l holds A[row = l & 31][k = 8 * (l >> 5) + e], e = 0..7   B: lane l holds B[k = 8 * (l >> 5) + e][col = l & 31]
//             C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// with A[i][k] = (k == i % 16), B[k][j] = 100 k + j (asymmetric)  =>  C[i][j] = 100 (i % 16) + j.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_bf16_layout mfma_bf16_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float *C) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; e++) {
        const int kk = 8 * h + e;
        a[e] = (__bf16)(kk == r % 16 ? 1.0f : 0.0f);
        b[e] = (__bf16)(float)(100 * kk + r);          // exact in bf16 only for small values: 100 k + j <= 1531 needs 11 bits -> use smaller
    }
    for (int e = 0; e < 8; e++) b[e] = (__bf16)(float)(8 * (8 * h + e) + (r & 7));      // <= 127: exact
    f32x16 c;
    for (int q = 0; q < 16; q++) c[q] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int q = 0; q < 16; q++) C[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = c[q];
}
int main() {
    float *d, h[1024];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) bad += h[i * 32 + j] != (float)(8 * (i % 16) + (j & 7));
    printf("mismatches under the hypothesised layout: %d of 1024\n", bad);
    if (bad) for (int i = 0; i < 4; i++) { for (int j = 0; j < 10; j++) printf("%6.0f", h[i * 32 + j]); printf("\n"); }
    return 0;
}
