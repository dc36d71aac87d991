// Microbenchmark 3: contiguous-segment f32 atomic row-update rate vs the popularity skew of the target
// rows: (a) Zipf(0.6) over 38k item rows + uniform (user-major schedule: Q[i], Q[j]);
// (b) Zipf(0.4) over 31.7k user rows + uniform over 38k item rows (item-major schedule: P[u], Q[j]);
// (c) all uniform.  2.5 M row updates each, 16 lanes x 64 B x 4 per row.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ __launch_bounds__(256) void k(float* tab, const int* rows, long n) {
  const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long t = wave * 4 + g; t < n; t += nw * 4) { float* p = tab + (long)rows[t] * 64; for (int e = 0; e < 4; e++) unsafeAtomicAdd(p + r + 16 * e, 1.0f); }
}
static std::vector<double> zipf_cdf(long n, double a) { std::vector<double> c(n); double s = 0; for (long i = 0; i < n; i++) { s += pow(i + 1.0, -a); c[i] = s; } return c; }
int main() {
  const long U = 31668, I = 38048, n = 2500000;
  std::mt19937_64 rng(1);
  auto cu = zipf_cdf(U, 0.4), ci = zipf_cdf(I, 0.6);
  std::vector<int> perm_u(U), perm_i(I); for (long i = 0; i < U; i++) perm_u[i] = i; for (long i = 0; i < I; i++) perm_i[i] = i;
  std::shuffle(perm_u.begin(), perm_u.end(), rng); std::shuffle(perm_i.begin(), perm_i.end(), rng);
  auto draw = [&](const std::vector<double>& c) { std::uniform_real_distribution<double> d(0, c.back()); return (long)(std::lower_bound(c.begin(), c.end(), d(rng)) - c.begin()); };
  float* tab; int* dr; hipMalloc(&tab, (U + I) * 256); hipMemset(tab, 0, (U + I) * 256); hipMalloc(&dr, n * 4);
  const char* names[3] = {"user-major: Zipf.6 items + uniform items", "item-major: Zipf.4 users + uniform items", "all uniform"};
  for (int cfg = 0; cfg < 3; cfg++) {
    std::vector<int> rows(n);
    for (long t = 0; t < n; t++) {
      if (t & 1) rows[t] = U + rng() % I;
      else if (cfg == 0) rows[t] = U + perm_i[draw(ci)];
      else if (cfg == 1) rows[t] = perm_u[draw(cu)];
      else rows[t] = rng() % (U + I);
    }
    hipMemcpy(dr, rows.data(), n * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 2048}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float best = 1e9;
      for (int rep = 0; rep < 6; rep++) { hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, dr, n); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms; }
      printf("%-44s blocks=%4d  %.3f ms  %.2f G rows/s\n", names[cfg], blocks, best, n / best / 1e6);
    }
  }
  return 0;
}
