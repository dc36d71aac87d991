// Microbenchmark 4 (round 3): the MEMORY side of one BPR epoch at the Yelp2018 shape under four schedules -- the row loads and
// the atomic row updates each schedule issues, no arithmetic (the SGD kernel is bound by the atomic units) -- to
// decide which schedule is worth a full kernel.  A row update = 16 lanes x 4 contiguous-segment f32 atomics (256 B), as in
// bpr_sgd.hip.  Triplets: u ~ Zipf(0.4) over 31,668 users, i ~ Zipf(0.6) over 38,048 items, j uniform; 1,252,669 of them.
//   user   user-major (round 1):  P[u] in registers along the user's run; per triplet atomics on Q[i], Q[j]
//   item   item-major (shipped):  Q[i] in registers along the item's run, flushed every 16; per triplet atomics on P[u], Q[j]
//   owner  owner-computes users:  a workgroup owns a block of users, their P rows live in LDS (ds_add); triplets of the block in
//          item order, Q[i] in registers along the (block, item) run; per triplet atomic on Q[j], per run on Q[i]
//   defer  deferred negatives:    pass A = item-major without the Q[j] atomic (its coefficient is logged, 4 B); pass B walks
//          the triplets in j order, re-reads P[u], keeps Q[j] in registers along the j run, one atomic per run
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
constexpr int LD = 64;
__device__ inline void row_atomic(float* tab, int row, int r, float v) { float* p = tab + (long)row * LD; for (int e = 0; e < 4; e++) unsafeAtomicAdd(p + r + 16 * e, v); }
__device__ inline float row_load(const float* tab, int row, int r) { const float* p = tab + (long)row * LD; float s = 0; for (int e = 0; e < 4; e++) s += p[r + 16 * e]; return s; }
// generic walker: a group of 16 lanes takes chunks of CH consecutive triplets of the given order.  a_row: the row kept in
// registers along its run (atomic at run end / every `flush`), b_row and c_row: loaded per triplet; b_atomic / c_atomic: updated per triplet
template <bool LOADS>
__global__ __launch_bounds__(256) void walk_t(float* A, float* B, float* C, const int* a_row, const int* b_row, const int* c_row, long n, int CH, int flush,
                                              int b_atomic, int c_atomic, float* glog, long stride) {
  const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
  const long grp = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + g, ngrp = (((long)gridDim.x * blockDim.x) >> 6) * 4;
  const long nch = (n + CH - 1) / CH;
  for (long s = grp; s < nch; s += ngrp) {
    const long c = (s * stride) % nch;
    const long t0 = c * CH, t1 = t0 + CH < n ? t0 + CH : n;
    int cur = a_row[t0], since = 0; float acc = LOADS ? row_load(A, cur, r) : 1.f;
    for (long t = t0; t < t1; t++) {
      const int a = a_row[t];
      if (a != cur || since == flush) { row_atomic(A, cur, r, acc * 1e-9f); cur = a; since = 0; if (LOADS) acc = row_load(A, cur, r); }
      const float vb = LOADS ? row_load(B, b_row[t], r) : 1.f, vc = (LOADS && c_row) ? row_load(C, c_row[t], r) : 1.f;
      acc += vb + vc; since++;
      if (b_atomic) row_atomic(B, b_row[t], r, vc * 1e-9f);
      if (c_atomic) row_atomic(C, c_row[t], r, vb * 1e-9f);
      if (glog && r == 0) glog[t] = vb;
    }
    row_atomic(A, cur, r, acc * 1e-9f);
  }
}
__global__ __launch_bounds__(256) void walk(float* A, float* B, float* C, const int* a_row, const int* b_row, const int* c_row, long n, int CH, int flush,
                                            int b_atomic, int c_atomic, float* glog, long stride) {
  const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
  const long grp = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + g, ngrp = (((long)gridDim.x * blockDim.x) >> 6) * 4;
  const long nch = (n + CH - 1) / CH;
  for (long s = grp; s < nch; s += ngrp) {
    const long c = (s * stride) % nch;
    const long t0 = c * CH, t1 = t0 + CH < n ? t0 + CH : n;
    int cur = a_row[t0], since = 0; float acc = row_load(A, cur, r);
    for (long t = t0; t < t1; t++) {
      const int a = a_row[t];
      if (a != cur || since == flush) { row_atomic(A, cur, r, acc * 1e-9f); cur = a; since = 0; acc = row_load(A, cur, r); }
      const float vb = row_load(B, b_row[t], r), vc = c_row ? row_load(C, c_row[t], r) : 0.f;
      acc += vb + vc; since++;
      if (b_atomic) row_atomic(B, b_row[t], r, vc * 1e-9f);
      if (c_atomic) row_atomic(C, c_row[t], r, vb * 1e-9f);
      if (glog && r == 0) glog[t] = vb;
    }
    row_atomic(A, cur, r, acc * 1e-9f);
  }
}
// owner-computes: block b owns users [ub[b], ub[b+1]) -- their rows in LDS -- and the triplets [tb[b], tb[b+1]) (item order)
__global__ __launch_bounds__(1024) void owner(float* P, float* Q, const int* u, const int* i, const int* j, const int* ub, const int* tb, int CH) {
  extern __shared__ float lds[];
  const int b = blockIdx.x, u0 = ub[b], nu = ub[b + 1] - u0;
  for (int k = threadIdx.x; k < nu * LD; k += blockDim.x) lds[k] = P[(long)u0 * LD + k];
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, grp = (threadIdx.x >> 6) * 4 + (lane >> 4), ngrp = (blockDim.x >> 6) * 4;
  const long t_lo = tb[b], t_hi = tb[b + 1], nch = (t_hi - t_lo + CH - 1) / CH;
  for (long c = grp; c < nch; c += ngrp) {
    const long t0 = t_lo + c * CH, t1 = t0 + CH < t_hi ? t0 + CH : t_hi;
    int cur = i[t0]; float acc = row_load(Q, cur, r);
    for (long t = t0; t < t1; t++) {
      const int a = i[t];
      if (a != cur) { row_atomic(Q, cur, r, acc * 1e-9f); cur = a; acc = row_load(Q, cur, r); }
      float* p = lds + (long)(u[t] - u0) * LD;
      float vp = 0; for (int e = 0; e < 4; e++) vp += p[r + 16 * e];
      const float vj = row_load(Q, j[t], r);
      acc += vp + vj;
      for (int e = 0; e < 4; e++) atomicAdd(p + r + 16 * e, vj * 1e-9f);       // ds_add_f32: several groups of the block may hold the same user
      row_atomic(Q, j[t], r, vp * 1e-9f);
    }
    row_atomic(Q, cur, r, acc * 1e-9f);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < nu * LD; k += blockDim.x) P[(long)u0 * LD + k] = lds[k];
}
static std::vector<double> zipf_cdf(long n, double a) { std::vector<double> c(n); double s = 0; for (long k = 0; k < n; k++) { s += pow(k + 1.0, -a); c[k] = s; } return c; }
template <typename F> float time_ms(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float best = 1e9; for (int rep = 0; rep < 6; rep++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms; } return best; }
int main() {
  const long U = 31668, I = 38048, n = 1252669;
  std::mt19937_64 rng(1);
  auto cu = zipf_cdf(U, 0.4), ci = zipf_cdf(I, 0.6);
  std::vector<int> pu(U), pi(I); std::iota(pu.begin(), pu.end(), 0); std::iota(pi.begin(), pi.end(), 0);
  std::shuffle(pu.begin(), pu.end(), rng); std::shuffle(pi.begin(), pi.end(), rng);
  auto draw = [&](const std::vector<double>& c) { std::uniform_real_distribution<double> d(0, c.back()); return (long)(std::lower_bound(c.begin(), c.end(), d(rng)) - c.begin()); };
  std::vector<int> u(n), i(n), j(n);
  for (long t = 0; t < n; t++) { u[t] = pu[draw(cu)]; i[t] = pi[draw(ci)]; j[t] = rng() % I; }
  auto sorted_by = [&](auto key) { std::vector<long> o(n); std::iota(o.begin(), o.end(), 0L); std::stable_sort(o.begin(), o.end(), [&](long a, long b) { return key(a) < key(b); }); return o; };
  auto take = [&](const std::vector<int>& v, const std::vector<long>& o) { std::vector<int> r(n); for (long t = 0; t < n; t++) r[t] = v[o[t]]; return r; };
  float *P, *Q, *glog; hipMalloc(&P, U * LD * 4); hipMalloc(&Q, I * LD * 4); hipMalloc(&glog, n * 4); hipMemset(P, 0, U * LD * 4); hipMemset(Q, 0, I * LD * 4);
  int *du, *di, *dj; hipMalloc(&du, n * 4); hipMalloc(&di, n * 4); hipMalloc(&dj, n * 4);
  auto up = [&](const std::vector<int>& a, const std::vector<int>& b, const std::vector<int>& c) { hipMemcpy(du, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(di, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dj, c.data(), n * 4, hipMemcpyHostToDevice); };
  const int CH = 34; int BLK = 256; const long nch = (n + CH - 1) / CH; long stride = (long)(nch * 0.6180339887); while (std::gcd(stride, nch) != 1) stride++;
  auto report = [&](const char* name, float ms, double atomics) { printf("%-58s %.3f ms   %.2f row-atomics/triplet   normalised %.2f of 8 TB/s\n", name, ms, atomics, 1548.0 * n / (ms * 1e-3) / 8e12); };
  // atomics per triplet of each schedule, counted on the host
  auto runs = [&](const std::vector<int>& key, int flush) { long a = 0; for (long c = 0; c < nch; c++) { long t0 = c * CH, t1 = std::min(n, t0 + CH); int cur = key[t0], since = 0; for (long t = t0; t < t1; t++) { if (key[t] != cur || since == flush) { a++; cur = key[t]; since = 0; } since++; } a++; } return (double)a; };
  { auto o = sorted_by([&](long t) { return u[t]; }); auto su = take(u, o), si = take(i, o), sj = take(j, o); up(su, si, sj);
    float ms = time_ms([&] { hipLaunchKernelGGL(walk, dim3(BLK), dim3(256), 0, 0, P, Q, Q, du, di, dj, n, CH, 1 << 30, 1, 1, (float*)nullptr, 1L); });
    report("user-major: P[u] run, atomics Q[i] + Q[j]", ms, 2.0 + runs(su, 1 << 30) / n); }
  auto oi = sorted_by([&](long t) { return i[t]; }); auto iu = take(u, oi), ii = take(i, oi), ij = take(j, oi);
  { up(iu, ii, ij);
    float ms = time_ms([&] { hipLaunchKernelGGL(walk, dim3(BLK), dim3(256), 0, 0, Q, P, Q, di, du, dj, n, CH, 16, 1, 1, (float*)nullptr, stride); });
    report("item-major (shipped): Q[i] run / 16, atomics P[u] + Q[j]", ms, 2.0 + runs(ii, 16) / n);
    for (int blocks : {256, 1024}) {
      float m0 = time_ms([&] { hipLaunchKernelGGL(walk_t<false>, dim3(blocks), dim3(256), 0, 0, Q, P, Q, di, du, dj, n, CH, 16, 1, 1, (float*)nullptr, stride); });
      char nm[96]; snprintf(nm, 96, "  the same stream, ATOMICS ONLY (no row loads), %d blocks", blocks); report(nm, m0, 2.0 + runs(ii, 16) / n);
      float m1 = time_ms([&] { hipLaunchKernelGGL(walk_t<true>, dim3(blocks), dim3(256), 0, 0, Q, P, Q, di, du, dj, n, CH, 16, 0, 0, (float*)nullptr, stride); });
      snprintf(nm, 96, "  the same stream, LOADS ONLY (run-end atomics kept), %d blocks", blocks); report(nm, m1, runs(ii, 16) / n);
    } }
  for (int B : {64, 128, 256}) {
    std::vector<int> ub(B + 1), blk_of(U); for (int b = 0; b <= B; b++) ub[b] = (int)((long)U * b / B); for (int b = 0; b < B; b++) for (int x = ub[b]; x < ub[b + 1]; x++) blk_of[x] = b;
    auto o = sorted_by([&](long t) { return (long)blk_of[u[t]] * I + i[t]; }); auto su = take(u, o), si = take(i, o), sj = take(j, o);
    std::vector<int> tb(B + 1, 0); for (long t = 0; t < n; t++) tb[blk_of[su[t]] + 1]++; for (int b = 0; b < B; b++) tb[b + 1] += tb[b];
    up(su, si, sj); int *dub, *dtb; hipMalloc(&dub, (B + 1) * 4); hipMalloc(&dtb, (B + 1) * 4); hipMemcpy(dub, ub.data(), (B + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(dtb, tb.data(), (B + 1) * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(U / B + 1) * LD * 4; hipFuncSetAttribute((const void*)owner, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double a = 0; for (int b = 0; b < B; b++) { for (long t0 = tb[b]; t0 < tb[b + 1]; t0 += CH) { long t1 = std::min<long>(tb[b + 1], t0 + CH); int cur = si[t0]; for (long t = t0; t < t1; t++) if (si[t] != cur) { a++; cur = si[t]; } a++; } }
    const int threads = B >= 256 ? 256 : 1024;
    float ms = time_ms([&] { hipLaunchKernelGGL(owner, dim3(B), dim3(threads), lds, 0, P, Q, du, di, dj, dub, dtb, CH); });
    char name[96]; snprintf(name, 96, "owner-computes users (%d blocks x %d thr): LDS P, atomics Q[i]-runs + Q[j]", B, threads); report(name, ms, 1.0 + a / n);
  }
  { up(iu, ii, ij);     // deferred negatives, whole epoch and 4 sub-epochs
    for (int BLK : {256, 1024})
    for (int S : {1, 4}) {
      const long per = (n + S - 1) / S; double atoms = 0; float total = 0;
      for (int s = 0; s < S; s++) {
        const long lo = s * per, cnt = std::min(n, lo + per) - lo; if (cnt <= 0) break;
        std::vector<int> subu(iu.begin() + lo, iu.begin() + lo + cnt), subi(ii.begin() + lo, ii.begin() + lo + cnt), subj(ij.begin() + lo, ij.begin() + lo + cnt);
        std::vector<long> oj(cnt); std::iota(oj.begin(), oj.end(), 0L); std::stable_sort(oj.begin(), oj.end(), [&](long a, long b) { return subj[a] < subj[b]; });
        std::vector<int> ju(cnt), jj(cnt); for (long t = 0; t < cnt; t++) { ju[t] = subu[oj[t]]; jj[t] = subj[oj[t]]; }
        int *dju, *djj; hipMalloc(&dju, cnt * 4); hipMalloc(&djj, cnt * 4); hipMemcpy(dju, ju.data(), cnt * 4, hipMemcpyHostToDevice); hipMemcpy(djj, jj.data(), cnt * 4, hipMemcpyHostToDevice);
        const long nchs = (cnt + CH - 1) / CH; long st = (long)(nchs * 0.6180339887); while (std::gcd(st, nchs) != 1) st++;
        total += time_ms([&] {
          hipLaunchKernelGGL(walk, dim3(BLK), dim3(256), 0, 0, Q, P, Q, di + lo, du + lo, dj + lo, cnt, CH, 16, 1, 0, glog, st);          // pass A
          hipLaunchKernelGGL(walk, dim3(BLK), dim3(256), 0, 0, Q, P, (float*)nullptr, djj, dju, (const int*)nullptr, cnt, CH, 1 << 30, 0, 0, (float*)nullptr, st);   // pass B
        });
        long a = 0; for (long c = 0; c < nchs; c++) { long t0 = c * CH, t1 = std::min(cnt, t0 + CH); int cur = jj[t0]; for (long t = t0; t < t1; t++) if (jj[t] != cur) { a++; cur = jj[t]; } a++; }
        long ai = 0; for (long c = 0; c < nchs; c++) { long t0 = c * CH, t1 = std::min(cnt, t0 + CH); int cur = subi[t0], since = 0; for (long t = t0; t < t1; t++) { if (subi[t] != cur || since == 16) { ai++; cur = subi[t]; since = 0; } since++; } ai++; }
        atoms += a + ai + cnt; hipFree(dju); hipFree(djj);
      }
      char name[128]; snprintf(name, 128, "deferred negatives, %d sub-epoch(s), %d blocks: pass A item-major + pass B j-runs", S, BLK); report(name, total, atoms / n);
    }
  }
  return 0;
}
