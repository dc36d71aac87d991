// Where does exact_walk_lds_kernel (the fused evaluation's exact per-user walk, qrec_amd/csrc/eval_topk.hip) spend its time?
// Phase stamps from the 100 MHz wall clock (staging / heapify / walk), update and group counts, for 1 / 5 / 256 users.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../qrec_amd/csrc -o exact_walk_probe exact_walk_probe.hip
#define QREC_WALK_PROBE
#include "../../qrec_amd/csrc/eval_topk.hip"
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>
#include <cmath>
namespace qrec { void set_error(const char *, ...) {} }     // the library's error sink (error.cpp), not linked here
// heapq on (score, id) tuples, as CPython runs it (Lib/heapq.py: _siftdown, _siftup, heapify, heapreplace)
typedef std::pair<float, int> Ent;
static void siftdown(std::vector<Ent> &h, int start, int pos) {
    const Ent item = h[pos];
    while (pos > start) { const int par = (pos - 1) >> 1; if (item < h[par]) { h[pos] = h[par]; pos = par; continue; } break; }
    h[pos] = item;
}
static void siftup(std::vector<Ent> &h, int pos) {
    const int end = (int)h.size(), start = pos;
    const Ent item = h[pos];
    int child = 2 * pos + 1;
    while (child < end) {
        const int right = child + 1;
        if (right < end && !(h[child] < h[right])) child = right;
        h[pos] = h[child]; pos = child; child = 2 * pos + 1;
    }
    h[pos] = item;
    siftdown(h, start, pos);
}
static std::vector<int> host_walk(const float *row, int n, int K) {
    std::vector<Ent> h;
    for (int i = 0; i < K; i++) h.push_back({row[i], i});
    for (int i = K / 2 - 1; i >= 0; i--) siftup(h, i);
    for (int i = K; i < n; i++) if (row[i] > h[0].first) { h[0] = {row[i], i}; siftup(h, 0); }
    std::stable_sort(h.begin(), h.end(), [](const Ent &a, const Ent &b) { return a.first > b.first; });
    std::vector<int> ids; for (auto &e : h) ids.push_back(e.second);
    return ids;
}
int main() {
    const int n_items = 38048, K = 20, max_users = 256;
    const int64_t row_len = (n_items + 31) / 32 * 32;
    std::vector<float> h((size_t)max_users * row_len);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.3f, 0.1f);
    for (auto &x : h) x = std::round(nd(rng) * 2048.f) / 2048.f;      // quantised: plenty of equal scores, the case the walk exists for
    float *rows, *sc; int32_t *ids;
    hipMalloc(&rows, h.size() * 4); hipMalloc(&sc, max_users * K * 4); hipMalloc(&ids, max_users * K * 4);
    hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)((n_items + 3) / 4 * 4) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&exact_walk_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int n : {1, 5, 5, 256}) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(exact_walk_lds_kernel, dim3(n), dim3(256), lds, 0, rows, row_len, n_items, K, ids, sc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static long long p[8 * 4096];
        hipMemcpyFromSymbol(p, HIP_SYMBOL(g_walk_probe), sizeof(p));
        printf("users %3d: %.1f us;  block 0: stage %.1f us, heapify %.1f us, walk %.1f us, %lld updates, %lld of %d groups entered\n", n, ms * 1e3,
               (p[1] - p[0]) / 100.0, (p[2] - p[1]) / 100.0, (p[3] - p[2]) / 100.0, p[4], p[5], (n_items - K + 511) / 512);
    }
    std::vector<int32_t> got((size_t)max_users * K);
    hipMemcpy(got.data(), ids, got.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int u = 0; u < max_users; u++) {
        const std::vector<int> want = host_walk(h.data() + (size_t)u * row_len, n_items, K);
        for (int a = 0; a < K; a++) bad += want[a] != got[(size_t)u * K + a];
    }
    printf("ids differing from heapq run on the host (256 users x %d): %d\n", K, bad);
    return bad != 0;
}
