// Microbenchmark 2: does the atomic rate depend on (a) allocation flavour, (b) XCD-affinity of
// rows (each row only ever touched from one XCD), and are same-address atomics from all XCDs exact?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// 16 lanes/row contiguous; MODE 0: rows[] as given; MODE 1: remap row so that row%8 == blockIdx%8 (XCD affinity)
template<int MODE>
__global__ __launch_bounds__(256) void k(float* tab, const int* rows, long n_updates, int n_rows) {
  const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int xcd = blockIdx.x & 7;
  for (long t = wave * 4 + g; t < n_updates; t += nwaves * 4) {
    int row = rows[t];
    if (MODE == 1) { row = (row & ~7) | xcd; if (row >= n_rows) row -= 8; }
    float* p = tab + (long)row * 64;
    for (int e = 0; e < 4; e++) unsafeAtomicAdd(p + r + 16 * e, 1.0f);
  }
}
__global__ void hammer(float* p, int iters) { for (int i = 0; i < iters; i++) unsafeAtomicAdd(p + (threadIdx.x & 63), 1.0f); }

template<int MODE> float run(float* tab, const int* rows, long n, int n_rows) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, tab, rows, n, n_rows); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 5; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, tab, rows, n, n_rows); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best;
}
int main() {
  const long n_rows = 38048, n_upd = 2500000;
  std::vector<int> rows(n_upd); std::mt19937_64 rng(1);
  for (long t = 0; t < n_upd; t++) rows[t] = rng() % n_rows;
  int* drows; CK(hipMalloc(&drows, n_upd * 4)); CK(hipMemcpy(drows, rows.data(), n_upd * 4, hipMemcpyHostToDevice));
  const char* names[4] = {"hipMalloc", "ExtMalloc uncached", "ExtMalloc finegrained", "hipMallocManaged"};
  for (int a = 0; a < 4; a++) {
    float* tab = nullptr; hipError_t e;
    size_t bytes = n_rows * 64 * 4;
    if (a == 0) e = hipMalloc(&tab, bytes);
    else if (a == 1) e = hipExtMallocWithFlags((void**)&tab, bytes, hipDeviceMallocUncached);
    else if (a == 2) e = hipExtMallocWithFlags((void**)&tab, bytes, hipDeviceMallocFinegrained);
    else e = hipMallocManaged(&tab, bytes);
    if (e != hipSuccess) { printf("%-22s alloc failed: %s\n", names[a], hipGetErrorString(e)); continue; }
    CK(hipMemset(tab, 0, bytes)); CK(hipDeviceSynchronize());
    float m0 = run<0>(tab, drows, n_upd, n_rows), m1 = run<1>(tab, drows, n_upd, n_rows);
    // exactness: 1024 blocks x 256 threads x 100 iters onto 64 floats -> each float gets 1024*4*100
    CK(hipMemset(tab, 0, 256)); hipLaunchKernelGGL(hammer, dim3(1024), dim3(256), 0, 0, tab, 100); CK(hipDeviceSynchronize());
    float h[64]; CK(hipMemcpy(h, tab, 256, hipMemcpyDeviceToHost));
    printf("%-22s any-XCD %.3f ms (%.2f G rows/s)  XCD-affine %.3f ms (%.2f G rows/s)  hammer sum=%.0f expect %d\n", names[a], m0, n_upd / m0 / 1e6, m1, n_upd / m1 / 1e6, h[5], 1024 * 4 * 100);
    hipFree(tab);
  }
  return 0;
}
