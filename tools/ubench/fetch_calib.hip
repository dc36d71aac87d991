// Calibration of rocprofv3 FETCH_SIZE for the hot kernel's read pattern: 4 B/lane loads, one
// contiguous 64-B segment per 16 lanes, random 256-B rows of a table far larger than the caches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void gather_rows(const float* tab, long n_rows, long n_reads, float* out) {
  const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float acc = 0;
  for (long t = wave * 4 + g; t < n_reads; t += nwaves * 4) {
    unsigned long long h = (unsigned long long)t * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    const float* p = tab + (long)(h % (unsigned long long)n_rows) * 64;
    for (int e = 0; e < 4; e++) acc += p[r + 16 * e];
  }
  if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void stream_f4(const float4* tab, long n4, float* out) {
  float acc = 0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (long)gridDim.x * blockDim.x) { float4 v = tab[k]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.f) out[0] = acc;
}
int main() {
  const long n_rows = 16L << 20;  // 16 Mi rows x 256 B = 4 GiB
  float* tab; float* out; hipMalloc(&tab, n_rows * 256); hipMalloc(&out, 4); hipMemset(tab, 0, n_rows * 256);
  const long n_reads = 8L << 20;   // 8 Mi row reads = 2 GiB expected
  hipLaunchKernelGGL(gather_rows, dim3(2048), dim3(256), 0, 0, tab, n_rows, n_reads, out);
  hipLaunchKernelGGL(stream_f4, dim3(2048), dim3(256), 0, 0, (const float4*)tab, (2L << 30) / 16, out);  // 2 GiB streaming
  hipDeviceSynchronize();
  printf("expected bytes: gather_rows %ld, stream_f4 %ld\n", n_reads * 256, 2L << 30);
  return 0;
}
