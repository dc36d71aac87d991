// Microbenchmark: f32/f64/u32/u64 global atomic-add throughput vs. coalescing pattern on
// random 256-B rows (development tool; results feed DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// each wave handles `per_wave` row-updates; rows[] random.
// MODE 0: 16 lanes/row, lane r adds elems 4r..4r+3 (4 instr, stride-16B)      [layout A]
// MODE 1: 16 lanes/row, lane r adds elems r+16e (4 instr, 64B contiguous)      [layout B]
// MODE 2: 64 lanes/row, 1 elem per lane (1 instr per row, 256 B contiguous)
// MODE 3: as 2 but f64 (2 rows of 32 doubles ... 64 lanes x 8B = 512B)
// MODE 4: as 2 but u32 integer add
// MODE 5: 32 lanes/row, u64 integer add covering 2 floats each (256 B per 32 lanes; 2 rows per instr)
// MODE 6: plain RMW float4 16 lanes/row (no atomics) for reference
// MODE 7: as 0 but with returning atomics (to see cost of return path)
template<int MODE>
__global__ __launch_bounds__(256) void k(float* tab, const int* rows, long n_updates, int ld) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  if (MODE == 0 || MODE == 1 || MODE == 6 || MODE == 7) {
    const int g = lane >> 4, r = lane & 15;
    for (long t = wave * 4 + g; t < n_updates; t += nwaves * 4) {
      float* p = tab + (long)rows[t] * ld;
      if (MODE == 0) { for (int e = 0; e < 4; e++) unsafeAtomicAdd(p + 4 * r + e, 1.0f); }
      else if (MODE == 1) { for (int e = 0; e < 4; e++) unsafeAtomicAdd(p + r + 16 * e, 1.0f); }
      else if (MODE == 7) { float acc = 0; for (int e = 0; e < 4; e++) acc += atomicAdd(p + 4 * r + e, 1.0f); if (acc == -1.f) p[0] = acc; }
      else { float4 v = *(float4*)(p + 4 * r); v.x += 1; v.y += 1; v.z += 1; v.w += 1; *(float4*)(p + 4 * r) = v; }
    }
  } else if (MODE == 2) {
    for (long t = wave; t < n_updates; t += nwaves) unsafeAtomicAdd(tab + (long)rows[t] * ld + lane, 1.0f);
  } else if (MODE == 3) {
    double* dt = (double*)tab;
    for (long t = wave * 2 + (lane >> 5); t < n_updates; t += nwaves * 2) unsafeAtomicAdd(dt + (long)rows[t] * (ld / 2) + (lane & 31), 1.0);
  } else if (MODE == 4) {
    unsigned* ut = (unsigned*)tab;
    for (long t = wave; t < n_updates; t += nwaves) atomicAdd(ut + (long)rows[t] * ld + lane, 1u);
  } else if (MODE == 5) {
    unsigned long long* ut = (unsigned long long*)tab;
    for (long t = wave * 2 + (lane >> 5); t < n_updates; t += nwaves * 2) atomicAdd(ut + (long)rows[t] * (ld / 2) + (lane & 31), 0x0000000100000001ull);
  }
}

template<int MODE> float run(float* tab, const int* rows, long n, int ld, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, rows, n, ld); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, rows, n, ld); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int ld = 64;
  for (int cfg = 0; cfg < 3; cfg++) {
    long n_rows = cfg == 0 ? 38048 : (cfg == 1 ? 70000 : 4000000);   // 9.7 MB, 17.9 MB, 1 GB
    long n_upd = cfg == 2 ? 20000000 : 2500000;
    bool zipf = cfg != 2;
    std::vector<int> rows(n_upd);
    std::mt19937_64 rng(1);
    if (zipf) { // half uniform (negatives), half zipf(0.6) (positives)
      std::vector<double> cdf(n_rows); double s = 0; for (long k = 0; k < n_rows; k++) { s += pow(k + 1.0, -0.6); cdf[k] = s; }
      std::uniform_real_distribution<double> U(0, s);
      for (long t = 0; t < n_upd; t++) { if (t & 1) rows[t] = rng() % n_rows; else { double x = U(rng); rows[t] = (int)(std::lower_bound(cdf.begin(), cdf.end(), x) - cdf.begin()); } }
    } else for (long t = 0; t < n_upd; t++) rows[t] = rng() % n_rows;
    float* tab; int* drows;
    CK(hipMalloc(&tab, n_rows * ld * 4)); CK(hipMemset(tab, 0, n_rows * ld * 4));
    CK(hipMalloc(&drows, n_upd * 4)); CK(hipMemcpy(drows, rows.data(), n_upd * 4, hipMemcpyHostToDevice));
    int blocks = 2048;
    float ms[8];
    ms[0] = run<0>(tab, drows, n_upd, ld, blocks); ms[1] = run<1>(tab, drows, n_upd, ld, blocks);
    ms[2] = run<2>(tab, drows, n_upd, ld, blocks); ms[3] = run<3>(tab, drows, n_upd, ld, blocks);
    ms[4] = run<4>(tab, drows, n_upd, ld, blocks); ms[5] = run<5>(tab, drows, n_upd, ld, blocks);
    ms[6] = run<6>(tab, drows, n_upd, ld, blocks); ms[7] = run<7>(tab, drows, n_upd, ld, blocks);
    const char* names[8] = {"f32 16l stride16B x4", "f32 16l contig64B x4", "f32 64l contig256B", "f64 32l contig256B", "u32 64l contig256B", "u64 32l contig256B", "plain RMW float4", "f32 returning 16l"};
    printf("cfg rows=%ld updates=%ld zipf=%d\n", n_rows, n_upd, (int)zipf);
    for (int m = 0; m < 8; m++) printf("  %-24s %8.3f ms  %7.2f G row-updates/s  %7.1f GB/s(256B rows)\n", names[m], ms[m], n_upd / ms[m] / 1e6, n_upd * 256.0 / ms[m] / 1e6);
    hipFree(tab); hipFree(drows);
  }
  return 0;
}
