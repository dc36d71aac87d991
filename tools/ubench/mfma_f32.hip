// micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 from registers (no memory traffic), by accumulator chains
// per wavefront and wavefronts per SIMD.  Peak by the spec sheet: 157.3 TFLOP/s = 64 cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) for (int q = 0; q < 16; q++) acc[c][q] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++) for (int q = 0; q < 16; q++) s += acc[c][q];
    if (s == 12345.678f) out[0] = s;
}
template <int CHAINS>
void run(int blocks_per_cu, float *d) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * 4 * iters * 16 * CHAINS, flop = n_mfma * 4096;
    printf("chains %d  blocks/CU %d (waves/SIMD %d)  %.3f ms  %.1f TFLOP/s  %.1f cycles/instr/SIMD at 2.4 GHz\n", CHAINS, blocks_per_cu, blocks_per_cu,
           ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_mfma / 1024));
}
int main() {
    float *d; hipMalloc(&d, 4);
    for (int bpc : {1, 2, 4}) { run<1>(bpc, d); run<2>(bpc, d); run<4>(bpc, d); }
    return 0;
}
