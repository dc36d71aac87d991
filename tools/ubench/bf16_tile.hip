// micro-benchmark behind the bf16 passes of the fused evaluation (sample_max_bf16_kernel / score_filter_bf16_kernel, round 6):
// what keeps "one 32-item tile x NU 32-user tiles, K = 64" -- 16 v_mfma_f32_32x32x16_bf16 (512 matrix-pipe cycles) + a max
// epilogue over the 64 accumulator registers -- at a third of the matrix pipe?
//   mode 0  MFMAs only, operands in registers, accumulators folded into one register at the very end
//   mode 1  + the per-tile epilogue: maximum of each user tile's 16 accumulators (v_max3 tree), compared with a running best
//   mode 2  + the item tile fetched from memory (fragment order: one contiguous KiB per load instruction), next tile prefetched
//   mode 3  mode 2 with the epilogue of tile t placed between the MFMA groups of tile t + 1 (two accumulator sets)
//   mode 4  mode 3 without loads
//   mode 5  mode 2 with sample_max_bf16_kernel's epilogue: the register number rides in the score's low 4 bits (16 v_and_or + the tree)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/bf16_tile tools/ubench/bf16_tile.hip ; run: tools/ubench/bf16_tile [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int NM = 4, NU = 4;

__device__ __forceinline__ float tile_max(const f32x16 &a) {
    const float x0 = fmaxf(fmaxf(a[0], a[1]), a[2]), x1 = fmaxf(fmaxf(a[3], a[4]), a[5]), x2 = fmaxf(fmaxf(a[6], a[7]), a[8]);
    const float x3 = fmaxf(fmaxf(a[9], a[10]), a[11]), x4 = fmaxf(fmaxf(a[12], a[13]), a[14]);
    return fmaxf(fmaxf(fmaxf(x0, x1), fmaxf(x2, x3)), fmaxf(x4, a[15]));
}

__device__ __forceinline__ float tagged_max(const f32x16 &a) {
    float t[16];
#pragma unroll
    for (int q = 0; q < 16; q++) t[q] = __uint_as_float((__float_as_uint(a[q]) & ~15u) | (unsigned)q);
    const float x0 = fmaxf(fmaxf(t[0], t[1]), t[2]), x1 = fmaxf(fmaxf(t[3], t[4]), t[5]), x2 = fmaxf(fmaxf(t[6], t[7]), t[8]);
    const float x3 = fmaxf(fmaxf(t[9], t[10]), t[11]), x4 = fmaxf(fmaxf(t[12], t[13]), t[14]);
    return fmaxf(fmaxf(fmaxf(x0, x1), fmaxf(x2, x3)), fmaxf(x4, t[15]));
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const __bf16 *__restrict__ Vb, int n_tiles, int tiles_per_wave, float *out, long long *cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 uu[NU][NM], va[NM], vb[NM];
    for (int kk = 0; kk < NU; kk++)
        for (int m = 0; m < NM; m++)
            for (int e = 0; e < 8; e++) uu[kk][m][e] = (__bf16)(0.01f * (float)((lane + 3 * kk + 5 * m + e) & 15));
    for (int m = 0; m < NM; m++)
        for (int e = 0; e < 8; e++) { va[m][e] = (__bf16)(0.02f * (float)((lane + m + e) & 7)); vb[m][e] = va[m][e]; }
    const int t0 = (blockIdx.x * 4 + wave) % n_tiles;
    auto load_tile = [&](int t, bf16x8 (&dst)[NM]) {
        const __bf16 *frag = Vb + (((long long)(t % n_tiles) * NM) * 64 + lane) * 8;
#pragma unroll
        for (int m = 0; m < NM; m++) dst[m] = *reinterpret_cast<const bf16x8 *>(frag + m * 512);
    };
    constexpr bool LOADS = (MODE == 2 || MODE == 3 || MODE == 5);
    if (LOADS) load_tile(t0, va);
    float best[NU];
    for (int kk = 0; kk < NU; kk++) best[kk] = -1e30f;
    f32x16 fold[NU];
    for (int kk = 0; kk < NU; kk++)
        for (int q = 0; q < 16; q++) fold[kk][q] = 0.f;
    const long long c0 = __builtin_readcyclecounter();
    if constexpr (MODE <= 2 || MODE == 5) {
        for (int it = 0; it < tiles_per_wave; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                bf16x8(&cur)[NM] = half ? vb : va;
                bf16x8(&nxt)[NM] = half ? va : vb;
                if (LOADS) load_tile(t0 + (it + half + 1) * 37, nxt);
                else cur[0][0] = (__bf16)(float)(it + half);     // not loop invariant
                f32x16 acc[NU];
#pragma unroll
                for (int kk = 0; kk < NU; kk++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[kk][q] = 0.f;
#pragma unroll
                for (int m = 0; m < NM; m++)
#pragma unroll
                    for (int kk = 0; kk < NU; kk++) acc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m], uu[kk][m], acc[kk], 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int kk = 0; kk < NU; kk++) fold[kk][(it + half) & 15] += acc[kk][(it + half) & 15];       // one register per tile: keeps the MFMAs alive
                } else {
#pragma unroll
                    for (int kk = 0; kk < NU; kk++) {
                        const float x = MODE == 5 ? tagged_max(acc[kk]) : tile_max(acc[kk]);
                        if (x > best[kk]) best[kk] = x;
                    }
                }
            }
        }
    } else {
        // the epilogue of tile i - 1 between the MFMA groups of tile i: accumulator sets A / B and operand sets va / vb by unrolling
        f32x16 accA[NU], accB[NU];
#pragma unroll
        for (int kk = 0; kk < NU; kk++)
#pragma unroll
            for (int q = 0; q < 16; q++) { accA[kk][q] = 0.f; accB[kk][q] = 0.f; }
        if (LOADS) load_tile(t0 + 37, vb);
#pragma unroll
        for (int m = 0; m < NM; m++)
#pragma unroll
            for (int kk = 0; kk < NU; kk++) accA[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[m], uu[kk][m], accA[kk], 0, 0, 0);
        for (int i = 1; i + 1 < tiles_per_wave; i += 2) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                bf16x8(&cur)[NM] = half ? va : vb;          // tile i + half
                bf16x8(&nxt)[NM] = half ? vb : va;          // tile i + half + 1: its registers fed tile i + half - 1, whose MFMAs are issued
                f32x16(&accN)[NU] = half ? accA : accB;     // being formed
                f32x16(&accO)[NU] = half ? accB : accA;     // finished one tile ago
                if (LOADS) load_tile(t0 + (i + half + 1) * 37, nxt);
                else cur[0][0] = (__bf16)(float)(i + half);
#pragma unroll
                for (int m = 0; m < NM; m++) {
#pragma unroll
                    for (int kk = 0; kk < NU; kk++) {
                        if (m == 0) {
                            f32x16 z;
#pragma unroll
                            for (int q = 0; q < 16; q++) z[q] = 0.f;
                            accN[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m], uu[kk][m], z, 0, 0, 0);
                        } else {
                            accN[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[m], uu[kk][m], accN[kk], 0, 0, 0);
                        }
                    }
                    // user tile m's epilogue of the PREVIOUS item tile rides under this group's four MFMAs
                    const float x = tile_max(accO[m]);
                    if (x > best[m]) best[m] = x;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < NU; kk++) { const float x = tile_max(accA[kk]) + tile_max(accB[kk]); if (x > best[kk]) best[kk] = x; }
    }
    const long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int kk = 0; kk < NU; kk++) {
        s += best[kk];
        for (int q = 0; q < 16; q++) s += fold[kk][q];
    }
    out[(blockIdx.x * 256 + threadIdx.x)] = s;
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = c1 - c0;
}

template <int MODE>
void run(const __bf16 *Vb, int n_tiles, int tiles_per_wave, int blocks, float *out, long long *cyc, std::vector<long long> &h) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, Vb, n_tiles, tiles_per_wave, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, Vb, n_tiles, tiles_per_wave, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks * 4; i++) avg += (double)h[i]; avg /= blocks * 4;
    const double mfmas = (double)blocks * 4 * tiles_per_wave * 16;
    const double flops = mfmas * 2.0 * 32 * 32 * 16;
    printf("mode %d  blocks %5d  %.3f ms  %.0f TFLOP/s (%.2f of 2500)  s_memtime ticks per tile per wave %.0f (100 MHz ticks x 24 = %.0f shader cycles; matrix pipe alone: 512)\n",
           MODE, blocks, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0, avg / tiles_per_wave, avg / tiles_per_wave * 24.0);
}

int main(int argc, char **argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int n_tiles = argc > 2 ? atoi(argv[2]) : 1189, tiles_per_wave = 512, blocks = 256 * per_cu;
    __bf16 *Vb; float *out; long long *cyc;
    hipMalloc(&Vb, (size_t)n_tiles * NM * 64 * 8 * 2);
    hipMemset(Vb, 0x3c, (size_t)n_tiles * NM * 64 * 8 * 2);
    hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    std::vector<long long> h(blocks * 4);
    run<0>(Vb, n_tiles, tiles_per_wave, blocks, out, cyc, h);
    run<1>(Vb, n_tiles, tiles_per_wave, blocks, out, cyc, h);
    run<2>(Vb, n_tiles, tiles_per_wave, blocks, out, cyc, h);
    run<3>(Vb, n_tiles, tiles_per_wave, blocks, out, cyc, h);
    run<4>(Vb, n_tiles, tiles_per_wave, blocks, out, cyc, h);
    run<5>(Vb, n_tiles, tiles_per_wave, blocks, out, cyc, h);
    return 0;
}
