// Full-rank evaluation: score every item for a batch of users, mask the user's rated items
// to 0, keep the N largest -- base/recommender.py:143-150, util/qmath.py:134-146.
//
//   (1) score_kernel   S_T[item][b] = V[item] . U[user_ids[b]]       MFMA (the one GEMM on
//       the path): v_mfma_f32_32x32x2_f32 for fp32 tables, v_mfma_f64_16x16x4_f64 for the
//       fp64 tables of the numpy-path models.  A wavefront keeps its 32 (16) users' U
//       fragment in registers and streams item tiles; V (<= tens of MB) is L2/MALL resident.
//       The block is written TRANSPOSED (item-major) so that step (3) reads coalesced.
//   (2) mask_kernel    S_T[item][b] = 0 for the user's rated train items -- to ZERO, not -inf:
//       the reference lets rated items compete with negative scores (recommender.py:147-149).
//   (3) heap_topk_kernel   one lane per user walks the items in id order and keeps the
//       reference's min-heap of (score, id) tuples -- CPython heapq's exact sift order,
//       strict '>' replacement, stable descending sort -- so ties resolve exactly as in the
//       reference (bit-exact ids).  Per user this is 1 load + 1 compare per item; ~N ln(I/N)
//       heap updates.
//
// Bytes: (1) writes I*B*s, reads V once per user tile from cache; (3) reads I*B*s.
// FLOPs: 2*B*I*d on the matrix pipe.
#include <cstdlib>

#include <type_traits>

#include "common.h"

using namespace qrec;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Operand fetch (both kernels): the tables are zero-padded to the row stride ld (a multiple of the slot width: 32
// floats / 16 doubles), so a lane reads ITS row's slot with unconditional 16-byte loads; rows past the end are
// clamped (their results are not stored).  The next item tile's operands are requested before the current tile's
// MFMAs.  (First version: one guarded dword load per element, 81 loads and 99 branches per 32 MFMAs.)
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ---- (1) fp32: 32 items x 32 users per MFMA tile, K consumed 2 slots x 32 columns per 64-chunk.
// lane l: row = l&31, slot h = l>>5 owns columns [64c + 32h, 64c + 32h + 32) of chunk c.
// (the dot product does not care which physical column sits in which MFMA k-slot as long as
//  A and B agree.)
__global__ __launch_bounds__(256) void score_kernel_f32(
    const float *__restrict__ U, const float *__restrict__ V, int ld, int n_items,
    const int32_t *__restrict__ user_ids, int n_b, int b_pad, int item_tiles_per_wave,
    float *__restrict__ S_T, int tile_stride, int64_t upair_stride, int64_t u_stride) {
    // tile_stride > 1 (threshold pass of the fused evaluation): block row 32*t + q is item 32*t*tile_stride + q, i.e.
    // only every tile_stride-th item tile is scored.
    // General address of (block row, user b): S_T[row * b_pad + (b / 64) * upair_stride + (b % 64) * u_stride].  u_stride = 1 in the
    // layouts below; b_pad = 1, u_stride = row length, upair_stride = 64 row lengths: USER-MAJOR rows (a user's scores contiguous:
    // what the exact per-user walk of a handful of flagged users wants -- its 64 loads per step are one 256-B access).
    // Layout: score of (block row, user b) at S_T[row * b_pad + (b / 64) * upair_stride + b % 64].  upair_stride = 64:
    // the plain [rows][b_pad] block.  b_pad = 64 with upair_stride = rows * 64: 64-user panels one after another -- a
    // user's column then strides 256 B and shares every cache line with 31 neighbours (the exact per-user walk).
    // One wavefront serves TWO 32-user tiles per item tile: the item operand (re-read from L2 by every user tile that
    // needs it -- 990 user tiles x 9.7 MB at the Yelp shape) is fetched half as often.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int upair = blockIdx.x;                       // 64 users
    const int b0 = upair * 64 + r, b1 = b0 + 32;
    const int64_t uid0 = user_ids[b0 < n_b ? b0 : n_b - 1], uid1 = user_ids[b1 < n_b ? b1 : n_b - 1];   // columns >= n_b are never read
    const int n_chunks = (ld + 63) / 64;                // 64 columns per chunk
    const int n_item_tiles = ((n_items + 31) / 32 + tile_stride - 1) / tile_stride;
    const int t_begin = (blockIdx.y * 4 + wave) * item_tiles_per_wave;
    int t_end = t_begin + item_tiles_per_wave;
    if (t_end > n_item_tiles) t_end = n_item_tiles;
    if (t_begin >= t_end) return;

    for (int c = 0; c < n_chunks; c++) {
        const int col0 = 64 * c + 32 * h;
        const bool kv = col0 < ld;                      // ld = 32 (mod 64): the last chunk's upper slot is empty
        const int kb = kv ? col0 : 0;
        const float keep = kv ? 1.f : 0.f;
        f32x4 ua[8], ub[8], va[8], vn[8];
        {
            const f32x4 *p0 = reinterpret_cast<const f32x4 *>(U + uid0 * ld + kb), *p1 = reinterpret_cast<const f32x4 *>(U + uid1 * ld + kb);
#pragma unroll
            for (int q = 0; q < 8; q++) { ua[q] = p0[q] * keep; ub[q] = p1[q] * keep; }
        }
        auto tile_ptr = [&](int t) {
            const int item = t * tile_stride * 32 + r;
            return reinterpret_cast<const f32x4 *>(V + (int64_t)(item < n_items ? item : n_items - 1) * ld + kb);
        };
        {
            const f32x4 *pv = tile_ptr(t_begin);
#pragma unroll
            for (int q = 0; q < 8; q++) va[q] = pv[q];
        }
        for (int t = t_begin; t < t_end; t++) {
            if (t + 1 < t_end) {
                const f32x4 *pv = tile_ptr(t + 1);
#pragma unroll
                for (int q = 0; q < 8; q++) vn[q] = pv[q];
            }
            f32x16 acc0, acc1;
            float *out = S_T + (int64_t)(t * 32) * b_pad + upair * upair_stride;
            const int item0 = t * tile_stride * 32;
            if (c == 0) {
#pragma unroll
                for (int q = 0; q < 16; q++) { acc0[q] = 0.f; acc1[q] = 0.f; }
            } else {  // accumulate across column chunks through the output block
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
                    const bool in = item0 + row < n_items;
                    acc0[q] = in ? out[(int64_t)row * b_pad + r * u_stride] : 0.f;
                    acc1[q] = in ? out[(int64_t)row * b_pad + (32 + r) * u_stride] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].x, ua[q].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].x, ub[q].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].y, ua[q].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].y, ub[q].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].z, ua[q].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].z, ub[q].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].w, ua[q].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[q].w, ub[q].w, acc1, 0, 0, 0);
            }
            // C/D: col = lane&31 (user), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (item)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
                if (item0 + row < n_items) {
                    out[(int64_t)row * b_pad + r * u_stride] = acc0[q];
                    out[(int64_t)row * b_pad + (32 + r) * u_stride] = acc1[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) va[q] = vn[q];
        }
    }
}

// ---- (1) fp64: 16 items x 16 users per MFMA tile, 4 slots x 16 columns per 64-chunk.
__global__ __launch_bounds__(256) void score_kernel_f64(
    const double *__restrict__ U, const double *__restrict__ V, int ld, int n_items,
    const int32_t *__restrict__ user_ids, int n_b, int b_pad, int item_tiles_per_wave,
    double *__restrict__ S_T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, h = lane >> 4;
    const int utile = blockIdx.x;                       // 16 users
    const int b = utile * 16 + r;
    const int64_t uid = user_ids[b < n_b ? b : n_b - 1];
    const int n_chunks = (ld + 63) / 64;
    const int n_item_tiles = (n_items + 15) / 16;
    const int t_begin = (blockIdx.y * 4 + wave) * item_tiles_per_wave;
    int t_end = t_begin + item_tiles_per_wave;
    if (t_end > n_item_tiles) t_end = n_item_tiles;
    if (t_begin >= t_end) return;

    for (int c = 0; c < n_chunks; c++) {
        const int col0 = 64 * c + 16 * h;
        const bool kv = col0 < ld;
        const int kb = kv ? col0 : 0;
        const double keep = kv ? 1.0 : 0.0;
        f64x2 ub[8], va[8], vn[8];
        {
            const f64x2 *pu = reinterpret_cast<const f64x2 *>(U + uid * ld + kb);
#pragma unroll
            for (int q = 0; q < 8; q++) ub[q] = pu[q] * keep;
        }
        auto tile_ptr = [&](int t) {
            const int item = t * 16 + r;
            return reinterpret_cast<const f64x2 *>(V + (int64_t)(item < n_items ? item : n_items - 1) * ld + kb);
        };
        {
            const f64x2 *pv = tile_ptr(t_begin);
#pragma unroll
            for (int q = 0; q < 8; q++) va[q] = pv[q];
        }
        for (int t = t_begin; t < t_end; t++) {
            if (t + 1 < t_end) {
                const f64x2 *pv = tile_ptr(t + 1);
#pragma unroll
                for (int q = 0; q < 8; q++) vn[q] = pv[q];
            }
            f64x4 acc;
            double *out = S_T + (int64_t)(t * 16) * b_pad + utile * 16;
            if (c == 0) {
                acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = h + 4 * q;
                    acc[q] = (t * 16 + row < n_items) ? out[(int64_t)row * b_pad + r] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[q].x, ub[q].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[q].y, ub[q].y, acc, 0, 0, 0);
            }
            // f64 C/D: col = lane&15 (user), row = (lane>>4) + 4*reg (item)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = h + 4 * q;
                if (t * 16 + row < n_items) out[(int64_t)row * b_pad + r] = acc[q];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) va[q] = vn[q];
        }
    }
}

// ---- (2) rated train items -> score 0
template <typename T>
__global__ __launch_bounds__(256) void mask_kernel(const int32_t *__restrict__ user_ids, int n_b,
                                                   const int64_t *__restrict__ rated_indptr,
                                                   const int32_t *__restrict__ rated_items, int b_pad,
                                                   T *__restrict__ S_T, int tile, int tile_stride, int64_t upair_stride, int64_t u_stride) {
    // one wavefront per user of the batch.  tile_stride > 1: the block holds every tile_stride-th item tile only.
    const int b = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (b >= n_b) return;
    const int uid = user_ids[b];
    const int64_t beg = rated_indptr[uid], end = rated_indptr[uid + 1];
    for (int64_t e = beg + (threadIdx.x & 63); e < end; e += 64) {
        const int it = rated_items[e], t = it / tile;
        if (t % tile_stride == 0) S_T[((int64_t)(t / tile_stride) * tile + it % tile) * b_pad + (b >> 6) * upair_stride + (b & 63) * u_stride] = T(0);
    }
}

// ---- (3) the reference's heap top-K, one lane per user ------------------------------------
constexpr int kHeapThreads = 64;

template <typename T>
struct Heap {  // column-per-thread layout in LDS: element k of thread t at [k*64 + t]
    T *s;
    int32_t *id;
    int t;
    __device__ T S(int k) const { return s[k * kHeapThreads + t]; }
    __device__ int32_t I(int k) const { return id[k * kHeapThreads + t]; }
    __device__ void set(int k, T sv, int32_t iv) { s[k * kHeapThreads + t] = sv; id[k * kHeapThreads + t] = iv; }
};

// Python tuple order on (score, id)
template <typename T>
__device__ inline bool tuple_lt(T sa, int32_t ia, T sb, int32_t ib) {
    return (sa < sb) || (sa == sb && ia < ib);
}

// heapq._siftdown(heap, startpos, pos)   (H: any heap view with S(k), I(k), set(k, score, id))
template <typename H>
__device__ inline void sift_down(H &hp, int startpos, int pos) {
    const auto ns = hp.S(pos); const int32_t ni = hp.I(pos);
    while (pos > startpos) {
        const int parent = (pos - 1) >> 1;
        const auto ps = hp.S(parent); const int32_t pi = hp.I(parent);
        if (tuple_lt(ns, ni, ps, pi)) { hp.set(pos, ps, pi); pos = parent; continue; }
        break;
    }
    hp.set(pos, ns, ni);
}
// heapq._siftup(heap, pos): bubble the smaller child up to a leaf, then sift the item down
template <typename H>
__device__ inline void sift_up(H &hp, int n, int pos) {
    const int startpos = pos;
    const auto ns = hp.S(pos); const int32_t ni = hp.I(pos);
    int child = 2 * pos + 1;
    while (child < n) {
        const int right = child + 1;
        if (right < n && !tuple_lt(hp.S(child), hp.I(child), hp.S(right), hp.I(right))) child = right;
        hp.set(pos, hp.S(child), hp.I(child));
        pos = child;
        child = 2 * pos + 1;
    }
    hp.set(pos, ns, ni);
    sift_down(hp, startpos, pos);
}

template <typename T>
__global__ __launch_bounds__(kHeapThreads) void heap_topk_kernel(const T *__restrict__ S_T, int n_items,
                                                                 int b_pad, int n_b, int K,
                                                                 int32_t *__restrict__ ids_out,
                                                                 T *__restrict__ scores_out,
                                                                 const int32_t *__restrict__ flags,
                                                                 const int32_t *__restrict__ n_flagged, int many) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *hs = reinterpret_cast<T *>(smem);
    int32_t *hi = reinterpret_cast<int32_t *>(smem + (size_t)K * kHeapThreads * sizeof(T));
    const int b = blockIdx.x * kHeapThreads + threadIdx.x;
    if (b >= n_b) return;
    // as the fallback of the sliced path: only when MANY users need the sequential emulation (then this coalesced
    // lane-per-user scan is the cheapest way), and only for those users
    if (flags && (*n_flagged <= many || !flags[b])) return;
    Heap<T> hp{hs, hi, (int)threadIdx.x};
    const T *col = S_T + b;
    const int k = K < n_items ? K : n_items;
    for (int t = 0; t < k; t++) hp.set(t, col[(int64_t)t * b_pad], t);
    for (int t = k / 2 - 1; t >= 0; t--) sift_up(hp, k, t);  // heapify
    T root = hp.S(0);
    int t = k;
    // main scan: 32 (fp32) / 16 (fp64) loads in flight per lane ahead of the (rarely taken) heap update -- there is
    // less than one wavefront per SIMD here (one lane per user), so latency is hidden by depth, not by occupancy
    constexpr int kAhead = sizeof(T) == 4 ? 32 : 16;
    for (; t + kAhead <= n_items; t += kAhead) {
        T v[kAhead];
#pragma unroll
        for (int q = 0; q < kAhead; q++) v[q] = col[(int64_t)(t + q) * b_pad];
#pragma unroll
        for (int q = 0; q < kAhead; q++) {
            if (v[q] > root) { hp.set(0, v[q], t + q); sift_up(hp, k, 0); root = hp.S(0); }
        }
    }
    for (; t < n_items; t++) {
        const T v = col[(int64_t)t * b_pad];
        if (v > root) { hp.set(0, v, t); sift_up(hp, k, 0); root = hp.S(0); }
    }
    // list.sort(key=score, reverse=True): stable insertion sort, descending
    for (int a = 1; a < k; a++) {
        const T xs = hp.S(a); const int32_t xi = hp.I(a);
        int c = a - 1;
        while (c >= 0 && hp.S(c) < xs) { hp.set(c + 1, hp.S(c), hp.I(c)); c--; }
        hp.set(c + 1, xs, xi);
    }
    for (int a = 0; a < k; a++) {
        ids_out[(int64_t)b * K + a] = hp.I(a);
        scores_out[(int64_t)b * K + a] = hp.S(a);
    }
    for (int a = k; a < K; a++) { ids_out[(int64_t)b * K + a] = -1; scores_out[(int64_t)b * K + a] = T(0); }
}

// ---- (3b) sliced top-N: the fast path ---------------------------------------------------------
// The lane-per-user scan above leaves half the chip idle (31,668 users = 495 wavefronts for 1,024 SIMDs) and is
// bound by the bytes one lane can keep in flight (0.95 TB/s).  The reference's heap result is history-dependent only
// through TIES: if a user's N+1 largest scores are pairwise distinct, find_k_largest returns exactly the N
// largest in descending order (every one of them enters the heap and stays; the stable sort has nothing to break).
// So: (a) group_max_kernel + threshold_kernel + filter_kernel -- two streaming passes: the (N+1)-th largest of 64
// per-range maxima is a threshold tau that the user's N+1 best all reach, and the second pass keeps the scores >= tau
// (tried before: a per-slice heap -- with 64 users per wavefront some lane updates its heap in almost every step of
// the warm-up, 6x the heap work of the single scan, 3.1 ms -- and the exact heap of a 2,048-item prefix as threshold,
// 1.5 ms of lane-per-user latency plus 400 survivors to merge); (b) merge_topk_kernel -- one
// lane per user builds the global top-(N+1) from the few dozen survivors (as a multiset of values
// it is exact whatever the ties), writes the N best, and FLAGS the user if two of the N+1 values are equal or a slice
// overflowed; (c) flagged users are redone with the exact
// sequential emulation: few of them -> one wavefront each (exact_wave_kernel: 64 items per step, ballot for the
// first item above the heap root, lane 0 runs heapq's sift), many -> the lane-per-user kernel above.
constexpr int kSlices = 16;
__host__ inline size_t score_block_bytes(size_t elem, int n_items, int n_b) {
    const size_t b_pad = ((size_t)n_b + 63) / 64 * 64, rows = ((size_t)n_items + 31) / 32 * 32;
    return rows * b_pad * elem;
}

// (a1) per-user maxima over kGroups disjoint item ranges (pure streaming, kGroups lanes per user), then tau[b] = the
//      M-th largest of them: M DIFFERENT items reach tau, so the user's M best scores all do -- a valid threshold, and a
//      tight one (about 1.5 M items of 38 k survive it).
constexpr int kGroups = 64;
template <typename T>
__global__ __launch_bounds__(256) void group_max_kernel(const T *__restrict__ S_T, int n_items, int b_pad, int n_b,
                                                        int items_per_group, T *__restrict__ gmax) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, grp = blockIdx.y;
    if (b >= n_b) return;
    const T *col = S_T + b;
    const int t0 = grp * items_per_group;
    int t1 = t0 + items_per_group;
    if (t1 > n_items) t1 = n_items;
    T best = col[(int64_t)t0 * b_pad];             // every group is non-empty (host guarantees)
    constexpr int kAhead = sizeof(T) == 4 ? 16 : 8;
    int t = t0 + 1;
    for (; t + kAhead <= t1; t += kAhead) {
        T v[kAhead];
#pragma unroll
        for (int q = 0; q < kAhead; q++) v[q] = col[(int64_t)(t + q) * b_pad];
#pragma unroll
        for (int q = 0; q < kAhead; q++) best = v[q] > best ? v[q] : best;
    }
    for (; t < t1; t++) { const T v = col[(int64_t)t * b_pad]; best = v > best ? v : best; }
    gmax[(int64_t)grp * b_pad + b] = best;
}
template <typename T>
__global__ __launch_bounds__(256) void fill_neg_inf_kernel(T *__restrict__ rows, int b_pad, int n_b) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_b) rows[(int64_t)blockIdx.y * b_pad + b] = -__builtin_huge_val();
}
template <typename T>
__global__ __launch_bounds__(256) void threshold_kernel(const T *__restrict__ gmax, int b_pad, int n_b, int M, T *__restrict__ tau) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_b) return;
    T v[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; g++) v[g] = gmax[(int64_t)g * b_pad + b];
    // M-th largest of kGroups values: M rounds of "take the maximum out"
    T th = T(0);
    for (int r = 0; r < M; r++) {
        int arg = 0;
        T best = v[0];
#pragma unroll
        for (int g = 1; g < kGroups; g++)
            if (v[g] > best) { best = v[g]; arg = g; }
        th = best;
#pragma unroll
        for (int g = 0; g < kGroups; g++)
            if (g == arg) v[g] = -__builtin_huge_val();
    }
    tau[b] = th;
}

// (a2) kSlices lanes per user stream the catalogue again: an item whose score reaches tau[b] is appended to the lane's
//      candidate rows (a handful per lane); more than kSliceCap of them (a plateau of ties at tau) overflows and sends
//      the user to the exact emulation.
constexpr int kSliceCap = 32;
template <typename T>
__global__ __launch_bounds__(256) void filter_kernel(const T *__restrict__ S_T, int n_items, int b_pad, int n_b,
                                                     int items_per_slice, const T *__restrict__ tau, T *__restrict__ cand_s,
                                                     int32_t *__restrict__ cand_i, int32_t *__restrict__ cand_n) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, slice = blockIdx.y;
    if (b >= n_b) return;
    const T *col = S_T + b;
    const T th = tau[b];
    const int t0 = slice * items_per_slice;
    int t1 = t0 + items_per_slice;
    if (t1 > n_items) t1 = n_items;
    const int64_t row0 = (int64_t)slice * kSliceCap;
    int cnt = 0;
    constexpr int kAhead = sizeof(T) == 4 ? 16 : 8;
    int t = t0;
    for (; t + kAhead <= t1; t += kAhead) {
        T v[kAhead];
#pragma unroll
        for (int q = 0; q < kAhead; q++) v[q] = col[(int64_t)(t + q) * b_pad];
#pragma unroll
        for (int q = 0; q < kAhead; q++)
            if (v[q] >= th) {
                if (cnt < kSliceCap) { cand_s[(row0 + cnt) * b_pad + b] = v[q]; cand_i[(row0 + cnt) * b_pad + b] = t + q; }
                cnt++;
            }
    }
    for (; t < t1; t++) {
        const T v = col[(int64_t)t * b_pad];
        if (v >= th) {
            if (cnt < kSliceCap) { cand_s[(row0 + cnt) * b_pad + b] = v; cand_i[(row0 + cnt) * b_pad + b] = t; }
            cnt++;
        }
    }
    cand_n[(int64_t)slice * b_pad + b] = cnt;
}

template <typename T>
__global__ __launch_bounds__(kHeapThreads) void merge_topk_kernel(const T *__restrict__ cand_s, const int32_t *__restrict__ cand_i,
                                                                  const int32_t *__restrict__ cand_n, int b_pad, int n_b, int K,
                                                                  int32_t *__restrict__ ids_out, T *__restrict__ scores_out,
                                                                  int32_t *__restrict__ flags, int32_t *__restrict__ n_flagged,
                                                                  int32_t *__restrict__ flagged_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = K + 1;
    T *hs = reinterpret_cast<T *>(smem);
    int32_t *hi = reinterpret_cast<int32_t *>(smem + (size_t)M * kHeapThreads * sizeof(T));
    const int b = blockIdx.x * kHeapThreads + threadIdx.x;
    if (b >= n_b) return;
    Heap<T> hp{hs, hi, (int)threadIdx.x};
    // at least M items reach tau, so (unless a slice overflowed) the survivors fill the heap
    int m = 0;
    T rs = T(0);
    int32_t ri = 0;
    bool tie = false;
    for (int sl = 0; sl < kSlices; sl++) {
        int n = cand_n[(int64_t)sl * b_pad + b];
        if (n > kSliceCap) { tie = true; n = kSliceCap; }          // overflow: a plateau at tau -> exact emulation
        const int64_t row0 = (int64_t)sl * kSliceCap;
        for (int c = 0; c < n; c++) {
            const T v = cand_s[(row0 + c) * b_pad + b];
            const int32_t id = cand_i[(row0 + c) * b_pad + b];
            if (m < M) {
                hp.set(m, v, id);
                if (++m == M) {
                    for (int a = M / 2 - 1; a >= 0; a--) sift_up(hp, M, a);
                    rs = hp.S(0); ri = hp.I(0);
                }
            } else if (tuple_lt(rs, ri, v, id)) {
                hp.set(0, v, id); sift_up(hp, M, 0); rs = hp.S(0); ri = hp.I(0);
            }
        }
    }
    if (m < M) {                                    // only possible after an overflow: pad, the user is flagged anyway
        tie = true;
        for (; m < M; m++) hp.set(m, -__builtin_huge_val(), -1);
    }
    for (int a = 1; a < M; a++) {                   // descending insertion sort (tuple order)
        const T xs = hp.S(a); const int32_t xi = hp.I(a);
        int c = a - 1;
        while (c >= 0 && tuple_lt(hp.S(c), hp.I(c), xs, xi)) { hp.set(c + 1, hp.S(c), hp.I(c)); c--; }
        hp.set(c + 1, xs, xi);
    }
    for (int a = 1; a < M; a++) tie |= hp.S(a) == hp.S(a - 1);
    for (int a = 0; a < K; a++) {
        ids_out[(int64_t)b * K + a] = hp.I(a);
        scores_out[(int64_t)b * K + a] = hp.S(a);
    }
    flags[b] = tie ? 1 : 0;
    if (tie) flagged_list[atomicAdd(n_flagged, 1)] = b;
}

// A heap of <= 64 entries spread over the lanes of one wavefront: entry k lives in lane k's registers and is read /
// read with v_readlane, written by a per-lane select (k is wave-uniform: every lane runs the same sift).  An update costs a few
// dozen cross-lane moves instead of lane 0 chasing LDS latencies behind a barrier.
template <typename T>
struct LaneHeap {
    T s;             // this lane's entry
    int32_t id;
    int lane;
    static __device__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(k)); }
    __device__ T S(int k) const {
        if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, rl(__builtin_bit_cast(int, s), k));
        else {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, s);
            const unsigned lo = (unsigned)rl((int)(unsigned)b, k), hi = (unsigned)rl((int)(unsigned)(b >> 32), k);
            return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
        }
    }
    __device__ int32_t I(int k) const { return rl(id, k); }
    __device__ void set(int k, T sv, int32_t iv) {          // k, sv, iv wave-uniform: lane k takes the value
        const bool me = lane == k;
        s = me ? sv : s; id = me ? iv : id;
    }
};

// rank of this lane's key among the 64 keys of the wavefront (how many are larger): 64 x (2 v_readlane, compare, add) -- no LDS
// round trips, no dependent shuffles
__device__ __forceinline__ int wave_rank_u64(unsigned long long key) {
    const int lo = (int)(unsigned)key, hi = (int)(unsigned)(key >> 32);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, j) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, j);
        rank += o > key ? 1 : 0;
    }
    return rank;
}
__device__ __forceinline__ unsigned long long wave_read_u64(unsigned long long key, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), s) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, s);
}

// The same heap for fp32 scores as ONE 64-bit key per lane: (order-preserving image of the score) << 32 | id, so Python's tuple
// order on (score, id) is the unsigned order of the keys.  heapq._siftup(heap, pos) -- bubble the smaller child up to a leaf, then
// sift the item back down -- nets out to an insertion along the smaller-child path: the path's entries grow downwards (the
// subtree below pos is a heap), so the item passes exactly those smaller than itself, each moving up one level, and lands where
// the next one is larger.  Every lane finds its smaller child at once (two 64-bit cross-lane reads), the path is five chained
// v_readlane, and how far the item travels is one ballot: no loop, no branch (the level-by-level loop over the same data cost
// 0.75 us per root replacement, mostly VALU <-> SALU hand-offs; tools/ubench/exact_walk_probe.hip).
struct KeyHeap {
    unsigned long long key;      // this lane's entry (lanes >= k: unused)
    int lane, k;
    static __device__ unsigned long long make(float s, int32_t id) {
        unsigned u = __float_as_uint(s + 0.f);                     // -0 -> +0: Python compares them equal
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        return ((unsigned long long)u << 32) | (unsigned)id;
    }
    static __device__ float score_of(unsigned long long kk) {
        unsigned u = (unsigned)(kk >> 32);
        u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        return __uint_as_float(u);
    }
    __device__ unsigned long long at(int i) const { return wave_read_u64(key, i); }
    __device__ void sift(int pos, unsigned long long item) {       // pos, item wave-uniform; heap[pos] is taken to hold `item`
        const int c1 = 2 * lane + 1, c2 = c1 + 1;
        const unsigned long long k1 = __shfl(key, c1 & 63, kWave), k2 = __shfl(key, c2 & 63, kWave);
        const bool right = c2 < k && !(k1 < k2);                   // heapq: the right child unless left < right
        const int sc = c1 < k ? (right ? c2 : c1) : lane;          // a leaf points at itself
        const unsigned long long ck = right ? k2 : k1;             // the smaller child's key
        // the smaller-child path from pos (k <= 64: at most five levels below the root); no branches, five chained v_readlane
        const int p0 = __builtin_amdgcn_readfirstlane(pos);
        const int p1 = __builtin_amdgcn_readlane(sc, p0), p2 = __builtin_amdgcn_readlane(sc, p1), p3 = __builtin_amdgcn_readlane(sc, p2);
        const int p4 = __builtin_amdgcn_readlane(sc, p3), p5 = __builtin_amdgcn_readlane(sc, p4);
        const int rel = (31 - __builtin_clz(lane + 1)) - (31 - __builtin_clz(p0 + 1));      // this lane's level below pos
        const int pl = rel == 0 ? p0 : rel == 1 ? p1 : rel == 2 ? p2 : rel == 3 ? p3 : rel == 4 ? p4 : rel == 5 ? p5 : -1;
        const bool on = pl == lane;                                // the path holds one node per level
        // the path's keys grow downwards: those below `item` are a prefix; each moves up one level, the item lands behind them
        const int f = __popcll(__ballot(on && rel >= 1 && key < item));
        key = on && rel < f ? ck : (on && rel == f ? item : key);
    }
};

// Exact sequential emulation for ONE user per wavefront (the users merge_topk_kernel flagged, when they are few).
// Same algorithm as heap_topk_kernel, 64 items per step: every lane holds one score, a ballot finds the first lane
// whose score beats the heap root, lane 0 performs heapq's replace in LDS, the root is re-read and the remaining
// lanes of the step are tested again.  The user's scores are b_pad*sizeof(T) bytes apart (64 cache lines per step):
// fine for a handful of users, 16x read amplification if used for all -- hence the `many` gate.
template <typename T, bool REG>      // REG: K <= 64, the heap lives in the wavefront's registers (LaneHeap); else in LDS
__global__ __launch_bounds__(64) void exact_wave_kernel(const T *__restrict__ S_T, int n_items, int b_pad, int K,
                                                        const int32_t *__restrict__ n_flagged, const int32_t *__restrict__ flagged_list,
                                                        int many, int32_t *__restrict__ ids_out, T *__restrict__ scores_out,
                                                        int64_t upair_stride, int64_t u_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // n_flagged == nullptr: every user of the block (one wavefront each) -- the fused route's fallback batch
    const int nf = n_flagged ? *n_flagged : (int)gridDim.x;
    if (nf > many || (int)blockIdx.x >= nf) return;
    const int b = n_flagged ? flagged_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
    T *hs = reinterpret_cast<T *>(smem);                                       // heap of this wavefront: [K] scores, [K] ids
    int32_t *hi = reinterpret_cast<int32_t *>(smem + (size_t)K * sizeof(T));
    struct WHeap {   // same interface as Heap<T>, one heap per block
        T *s; int32_t *id;
        __device__ T S(int k) const { return s[k]; }
        __device__ int32_t I(int k) const { return id[k]; }
        __device__ void set(int k, T sv, int32_t iv) { s[k] = sv; id[k] = iv; }
    };
    typename std::conditional<REG, LaneHeap<T>, WHeap>::type hp;
    constexpr bool KEYED = REG && sizeof(T) == 4;                 // fp32, K <= 64: the heap as one 64-bit key per lane
    KeyHeap kh;
    const T *col = S_T + (b >> 6) * upair_stride + (b & 63) * u_stride;
    const int k = K < n_items ? K : n_items;
    if constexpr (KEYED) {
        kh.lane = lane; kh.k = k;
        kh.key = lane < k ? KeyHeap::make((float)col[(int64_t)lane * b_pad], lane) : ~0ull;
        for (int t = k / 2 - 1; t >= 0; t--) kh.sift(t, kh.at(t));                 // heapq.heapify
    } else if constexpr (REG) {
        hp.s = lane < k ? col[(int64_t)lane * b_pad] : T(0); hp.id = lane; hp.lane = lane;
        for (int t = k / 2 - 1; t >= 0; t--) sift_up(hp, k, t);
    } else {
        hp.s = hs; hp.id = hi;
        for (int t = lane; t < k; t += 64) hp.set(t, col[(int64_t)t * b_pad], t);
        __syncthreads();
        if (lane == 0)
            for (int t = k / 2 - 1; t >= 0; t--) sift_up(hp, k, t);
        __syncthreads();
    }
    T root;
    if constexpr (KEYED) root = (T)KeyHeap::score_of(kh.at(0)); else root = hp.S(0);
    // One wavefront per user and nothing else on the SIMD: the walk is a chain of memory round trips unless the loads of many
    // steps are in flight.  Steps go in groups of kGroup: the next group's kGroup loads per lane are issued before the
    // current group is examined (8 steps ahead, shifted through registers, measured 0.65 us per step = 0.39 ms per user).
    constexpr int kGroup = 32;
    T cur[kGroup], nxt[kGroup];
    auto fetch = [&](int t0, T (&dst)[kGroup]) {
#pragma unroll
        for (int q = 0; q < kGroup; q++) { const int t = t0 + 64 * q + lane; dst[q] = t < n_items ? col[(int64_t)t * b_pad] : T(0); }
    };
    fetch(k, cur);
    for (int g0 = k; g0 < n_items; g0 += 64 * kGroup) {
        if (g0 + 64 * kGroup < n_items) fetch(g0 + 64 * kGroup, nxt);
#pragma unroll
        for (int q = 0; q < kGroup; q++) {
            const int t0 = g0 + 64 * q, t = t0 + lane;
            const T v = cur[q];
            bool live = t < n_items;
            while (true) {
                const unsigned long long mask = __ballot(live && v > root);
                if (!mask) break;
                const int first = __builtin_ctzll(mask);
                const T fv = __shfl(v, first, kWave);
                if constexpr (KEYED) {
                    kh.sift(0, KeyHeap::make((float)fv, t0 + first));                  // heapq.heapreplace
                    root = (T)KeyHeap::score_of(kh.at(0));
                } else if constexpr (REG) {
                    hp.set(0, fv, t0 + first); sift_up(hp, k, 0);
                    root = hp.S(0);
                } else {
                    if (lane == 0) { hp.set(0, fv, t0 + first); sift_up(hp, k, 0); }
                    __syncthreads();
                    root = hp.S(0);
                }
                live = live && lane > first;
            }
        }
#pragma unroll
        for (int q = 0; q < kGroup; q++) cur[q] = nxt[q];
    }
    if constexpr (KEYED) {
        // list.sort(key=score, reverse=True), stable: entry a's rank = how many entries beat it or tie it from the left
        const float mine = KeyHeap::score_of(kh.key);
        int rank = 0;
        for (int c = 0; c < k; c++) {
            const float cs = KeyHeap::score_of(kh.at(c));
            rank += (cs > mine || (cs == mine && c < lane)) ? 1 : 0;
        }
        if (lane < k) { ids_out[(int64_t)b * K + rank] = (int32_t)(unsigned)(kh.key & 0xffffffffu); scores_out[(int64_t)b * K + rank] = (T)mine; }
        for (int a = k + lane; a < K; a += 64) { ids_out[(int64_t)b * K + a] = -1; scores_out[(int64_t)b * K + a] = T(0); }
    } else if constexpr (REG) {
        // list.sort(key=score, reverse=True), stable: entry a's rank = how many entries beat it or tie it from the left
        int rank = 0;
        for (int c = 0; c < k; c++) {
            const T cs = hp.S(c);
            rank += (cs > hp.s || (cs == hp.s && c < lane)) ? 1 : 0;
        }
        if (lane < k) { ids_out[(int64_t)b * K + rank] = hp.id; scores_out[(int64_t)b * K + rank] = hp.s; }
        for (int a = k + lane; a < K; a += 64) { ids_out[(int64_t)b * K + a] = -1; scores_out[(int64_t)b * K + a] = T(0); }
    } else {
        __syncthreads();
        if (lane == 0) {
            for (int a = 1; a < k; a++) {               // list.sort(key=score, reverse=True): stable, descending
                const T xs = hp.S(a); const int32_t xi = hp.I(a);
                int c = a - 1;
                while (c >= 0 && hp.S(c) < xs) { hp.set(c + 1, hp.S(c), hp.I(c)); c--; }
                hp.set(c + 1, xs, xi);
            }
        }
        __syncthreads();
        for (int a = lane; a < K; a += 64) {
            ids_out[(int64_t)b * K + a] = a < k ? hp.I(a) : -1;
            scores_out[(int64_t)b * K + a] = a < k ? hp.S(a) : T(0);
        }
    }
}

// The exact walk for the fused route's flagged users (fp32, K <= 64, user-major rows): a block per user stages the user's row in
// LDS -- 156 KB at a time, all four wavefronts copying with 16-byte accesses, dozens of loads in flight (5 us) -- and its first
// wavefront walks it from there, eight steps tested at a time: late in the row almost no item beats the heap's root.
// Measured at 38,048 items (tools/ubench/exact_walk_probe.hip): exact_wave_kernel 154 us whether a step's 64 loads were 64 cache
// lines or one; this kernel with the level-by-level sift 130 us (walk 119 us for 160 root replacements); with KeyHeap's
// branch-free sift 100 us -- what is left is one wavefront's dependent instruction chain, ~0.3 us per replacement.
constexpr int kWalkChunk = 39936;      // items per LDS stage
#ifdef QREC_WALK_PROBE                 // tools/ubench/exact_walk_probe.hip: phase stamps (100 MHz wall clock) and counts per block
__device__ long long g_walk_probe[8 * 4096];
#define QREC_WP(slot, val) do { if (threadIdx.x == 0) g_walk_probe[blockIdx.x * 8 + (slot)] = (val); } while (0)
#else
#define QREC_WP(slot, val) do { } while (0)
#endif
__global__ __launch_bounds__(256) void exact_walk_lds_kernel(const float *__restrict__ rows, int64_t row_len, int n_items, int K,
                                                             int32_t *__restrict__ ids_out, float *__restrict__ scores_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *stage = reinterpret_cast<float *>(smem);
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *row = rows + (int64_t)b * row_len;               // row_len: a multiple of 32, rows 128-byte aligned
    const int k = K < n_items ? K : n_items;
    KeyHeap kh;
    kh.lane = lane; kh.k = k; kh.key = ~0ull;
    float root = 0.f;
    [[maybe_unused]] long long n_updates = 0, n_groups_hit = 0;
    QREC_WP(0, (long long)wall_clock64());
    for (int c0 = 0; c0 < n_items; c0 += kWalkChunk) {
        const int n = n_items - c0 < kWalkChunk ? n_items - c0 : kWalkChunk;
        __syncthreads();                                          // the previous stage has been walked
        for (int i = threadIdx.x * 4; i < n; i += 1024)           // the row is padded to whole tiles: the last vector may run into the pad
            *reinterpret_cast<f32x4 *>(stage + i) = *reinterpret_cast<const f32x4 *>(row + c0 + i);
        __syncthreads();
        if (wave != 0) continue;
        if (c0 == 0) QREC_WP(1, (long long)wall_clock64());
        int i0 = 0;
        if (c0 == 0) {                                            // heapq.heapify of the first k items (k <= 64 <= n)
            kh.key = lane < k ? KeyHeap::make(stage[lane], lane) : ~0ull;
            for (int t = k / 2 - 1; t >= 0; t--) kh.sift(t, kh.at(t));
            root = KeyHeap::score_of(kh.at(0));
            i0 = k;
            QREC_WP(2, (long long)wall_clock64());
        }
        constexpr int kAhead = 8;
        for (; i0 < n; i0 += 64 * kAhead) {
            float v[kAhead];
            bool any = false;
#pragma unroll
            for (int q = 0; q < kAhead; q++) {
                const int i = i0 + 64 * q + lane;
                v[q] = i < n ? stage[i] : -__builtin_huge_valf();
                any |= v[q] > root;
            }
            if (!__any(any)) continue;
#ifdef QREC_WALK_PROBE
            n_groups_hit++;
#endif
#pragma unroll
            for (int q = 0; q < kAhead; q++) {
                const int t0 = c0 + i0 + 64 * q;
                bool live = true;
                while (true) {
                    const unsigned long long mask = __ballot(live && v[q] > root);
                    if (!mask) break;
                    const int first = __builtin_ctzll(mask);
                    const float fv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[q]), first));
                    kh.sift(0, KeyHeap::make(fv, t0 + first));                     // heapq.heapreplace
                    root = KeyHeap::score_of(kh.at(0));
                    live = live && lane > first;
#ifdef QREC_WALK_PROBE
                    n_updates++;
#endif
                }
            }
        }
    }
    if (wave != 0) return;
    QREC_WP(3, (long long)wall_clock64()); QREC_WP(4, n_updates); QREC_WP(5, n_groups_hit);
    // list.sort(key=score, reverse=True), stable: entry a's rank = how many entries beat it or tie it from the left
    const float mine = KeyHeap::score_of(kh.key);
    int rank = 0;
    for (int c = 0; c < k; c++) {
        const float cs = KeyHeap::score_of(kh.at(c));
        rank += (cs > mine || (cs == mine && c < lane)) ? 1 : 0;
    }
    if (lane < k) { ids_out[(int64_t)b * K + rank] = (int32_t)(unsigned)(kh.key & 0xffffffffu); scores_out[(int64_t)b * K + rank] = mine; }
    for (int a = k + lane; a < K; a += 64) { ids_out[(int64_t)b * K + a] = -1; scores_out[(int64_t)b * K + a] = 0.f; }
}

// ---- fused evaluation (fp32): score -> mask -> threshold filter in one pass, no B x I block ---------------------------
// base/recommender.py:143-150 + util/qmath.py:134-146 again, for the common case that a user's N+1 best scores are
// pairwise distinct (then find_k_largest returns exactly the N best, descending):
//   (A) threshold: score every kSampleStride-th item tile only (a block 1/8 the size), mask it, and take tau[b] = the
//       (N+1)-th largest of 64 range maxima of that sample -- N+1 DIFFERENT items reach tau, so the user's N+1 best do;
//       about (N+1) x kSampleStride x 1.1 items of the whole catalogue reach it;
//   (B) score_filter2_kernel_f32: the scoring kernel with the store replaced by a compare against tau in the MFMA
//       accumulators; an item that reaches tau is appended to the lane's private candidate list.  Nothing else is
//       written: the kernel is bound by the matrix pipe.  (Rated items are NOT looked up here: a bisection per hit in
//       the epilogue -- six dependent loads under divergence -- cost 2x the tile's MFMA time, measured 3.6 ms.)
//   (C) select_topk_kernel: one wavefront per user gathers the user's candidates (a few hundred), drops the user's
//       rated items (they count as 0 < tau; bisection, all lanes in parallel) and extracts the N+1 largest by repeated
//       wave-wide maximum; equal scores among them, an overflowed list or tau <= 0 (rated items, masked to 0, would
//       compete) flag the user;
//   (D) flagged users -- the cases where the heap's history matters -- go through the block path above (MFMA scores of
//       THOSE users, mask, the exact heap emulation), a few hundred at a time, and their rows are patched in.
constexpr int kSampleStride = 8;
constexpr int kListCap = 48;        // candidates per lane-private list (expected ~10)

// two 32-user tiles per wavefront (two interleaved accumulator chains); OCC = wavefronts per SIMD the registers are capped for
template <int NC, int OCC>
__global__ __launch_bounds__(256, OCC) void score_filter2_kernel_f32(
    const float *__restrict__ U, const float *__restrict__ V, int ld, int n_items, const int32_t *__restrict__ user_ids,
    int n_b, int item_tiles_per_wave, const float *__restrict__ tau, int n_lists, float *__restrict__ cand_s,
    int32_t *__restrict__ cand_i, int32_t *__restrict__ cand_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int upair = blockIdx.x;
    const int b0 = upair * 64 + r, b1 = b0 + 32;
    const int list = (blockIdx.y * 4 + wave) * 2 + h;
    // wavefront w of the user pair takes item tiles w, w + W, w + 2W, ... (W = wavefronts per user pair): every candidate
    // list samples the whole catalogue, so ids that correlate with popularity cannot overflow one list
    const int n_item_tiles = (n_items + 31) / 32;
    const int t_step = gridDim.y * 4;
    const int t_begin = blockIdx.y * 4 + wave, t_end = n_item_tiles;
    (void)item_tiles_per_wave;
    const bool live0 = b0 < n_b, live1 = b1 < n_b;
    const int64_t uid0 = user_ids[live0 ? b0 : n_b - 1], uid1 = user_ids[live1 ? b1 : n_b - 1];
    int cnt0 = 0, cnt1 = 0;
    if (t_begin < t_end) {
        const float th0 = live0 ? tau[b0] : __builtin_huge_valf(), th1 = live1 ? tau[b1] : __builtin_huge_valf();
        f32x4 ua[NC][8], ub[NC][8];
        int kb[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int col0 = 64 * c + 32 * h;
            const bool kv = col0 < ld;
            kb[c] = kv ? col0 : 0;
            const float keep = kv ? 1.f : 0.f;
            const f32x4 *p0 = reinterpret_cast<const f32x4 *>(U + uid0 * ld + kb[c]), *p1 = reinterpret_cast<const f32x4 *>(U + uid1 * ld + kb[c]);
#pragma unroll
            for (int q = 0; q < 8; q++) { ua[c][q] = p0[q] * keep; ub[c][q] = p1[q] * keep; }
        }
        auto load_tile = [&](int t, f32x4 (&dst)[NC][8]) {
            const int item = t * 32 + r;
            const float *row = V + (int64_t)(item < n_items ? item : n_items - 1) * ld;
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int q = 0; q < 8; q++) dst[c][q] = reinterpret_cast<const f32x4 *>(row + kb[c])[q];
        };
        float *cs0 = cand_s + ((int64_t)b0 * n_lists + list) * kListCap, *cs1 = cand_s + ((int64_t)b1 * n_lists + list) * kListCap;
        int32_t *ci0 = cand_i + ((int64_t)b0 * n_lists + list) * kListCap, *ci1 = cand_i + ((int64_t)b1 * n_lists + list) * kListCap;
        // one item tile: 64 MFMAs, then the accumulators against the two thresholds.  Every executed VALU instruction of
        // the epilogue costs matrix-pipe time (measured), so: the element tests are entered only when some lane's
        // 16-element maximum reaches its threshold, the item-range test exists only in the catalogue's last tile.
        auto do_tile = [&](int t, const f32x4 (&v)[NC][8]) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; q++) { acc0[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].x, ua[c][q].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].x, ub[c][q].x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].y, ua[c][q].y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].y, ub[c][q].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].z, ua[c][q].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].z, ub[c][q].z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].w, ua[c][q].w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].w, ub[c][q].w, acc1, 0, 0, 0);
                }
            // C/D: col = lane&31 (user), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (item); items visited in ascending id per lane
            const int item_base = t * 32 + 4 * h;
            if (t * 32 + 32 > n_items) {              // the last, partial tile: rows past the catalogue repeat its last item
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (item_base + (q & 3) + 8 * (q >> 2) >= n_items) { acc0[q] = -__builtin_huge_valf(); acc1[q] = -__builtin_huge_valf(); }
            }
            float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
            for (int q = 1; q < 16; q++) { m0 = fmaxf(m0, acc0[q]); m1 = fmaxf(m1, acc1[q]); }
            if (m0 >= th0) {
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (acc0[q] >= th0) {
                        if (cnt0 < kListCap) { cs0[cnt0] = acc0[q]; ci0[cnt0] = item_base + (q & 3) + 8 * (q >> 2); }
                        cnt0++;
                    }
            }
            if (m1 >= th1) {
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (acc1[q] >= th1) {
                        if (cnt1 < kListCap) { cs1[cnt1] = acc1[q]; ci1[cnt1] = item_base + (q & 3) + 8 * (q >> 2); }
                        cnt1++;
                    }
            }
        };
        // two operand buffers, used alternately (no register copies): tile t from one while tile t + t_step loads into the other
        f32x4 va[NC][8], vb[NC][8];
        load_tile(t_begin, va);
        int t = t_begin;
        while (true) {
            load_tile(t + t_step < t_end ? t + t_step : t_end - 1, vb);
            do_tile(t, va);
            t += t_step;
            if (t >= t_end) break;
            load_tile(t + t_step < t_end ? t + t_step : t_end - 1, va);
            do_tile(t, vb);
            t += t_step;
            if (t >= t_end) break;
        }
    }
    if (live0) cand_n[(int64_t)b0 * n_lists + list] = cnt0;
    if (live1) cand_n[(int64_t)b1 * n_lists + list] = cnt1;
}

// (C) one wavefront per user: gather the candidates of the user's lists into LDS, take the K+1 largest by repeated
// wave-wide maximum of (score, lower id first); flag ties among them / overflow / tau <= 0.
// ---- (B') the scoring pass as a bf16 FILTER ------------------------------------------------------------------------------
// The pass only has to find every item whose fp32 score reaches tau; the scores themselves are re-formed afterwards (select
// kernel, the SAME fp32 MFMA sequence as score_kernel_f32, so ids and scores stay bit-identical to the block route).  So it
// runs on bf16 copies of the two tables -- v_mfma_f32_32x32x16_bf16: 1/16 of the matrix-pipe time of the f32 form, half the
// operand bytes -- against a threshold lowered by a bound on what the rounding can cost a score:
//   u^ = u (1 + d), |d| <= 2^-8 (round to nearest, 8-bit significand), same for v  =>  |u^.v^ - u.v| <= (2^-7 + 2^-16) sum |u_k v_k|
//   <= (2^-7 + 2^-16) |u| |v|;  the fp32 accumulation inside the MFMAs adds < 64 * 2^-24 of the same sum.
// eps[b] = 2^-7 * 1.02 * |u_b| * max_items |v| -- every item with u.v >= tau has u^.v^ >= tau - eps.
// The budget behind the 2 %, in units of |u| |v|: the bound proper is 2^-7 = 7.8125e-3 and the slack 0.02 * 2^-7 = 1.56e-4; it
// has to hold (i) the cross term 2^-16 = 1.5e-5, (ii) the fp32 accumulation inside and between the MFMAs, d/16 steps of
// 2^-24 each on the running sum: <= 128 * 2^-24 = 7.6e-6 up to d = 128 -- whatever order the matrix pipe adds in, each partial
// sum is bounded by sum |u^_k v^_k|, (iii) the candidate lists' 4-bit score tag (selection compares re-formed fp32 scores, the
// tag only orders the bf16 ones inside a list: 15 ulp of fp32 = 1.8e-6 relative), (iv) the norms' own rounding, covered by their
// factor 1.0001 (to_bf16_kernel).  Sum ~ 2.5e-5, a sixth of the slack.  tests/test_eval_bf16_bound.py checks the inequality
// numerically on the CPU; tests/test_gpu_eval.py::test_bf16_filter_is_complete_on_adversarial_tables holds the hardware's own
// accumulation to it (item and user norms over six decades, cancelling coordinates, clusters of near-ties at the threshold).
// Non-finite tables (a diverged run -- the reference stops such a run at the loss check, iterativeRecommender.py:84-86, before
// any evaluation): a norm is inf/NaN, tau_low = tau - inf/NaN admits nothing or everything, the user's lists come back empty or
// overflowed and the select kernel flags the user for the exact walk -- slow, not wrong.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// rows [rows][ld] fp32 (row_ids: gather, may be null) -> bf16 copy [rows][ld]; norm[row] = |x| (fp32); pad rows (>= rows) zero.
// TILED (the item table): the copy is laid out in the order the MFMA loops fetch it -- 32-row tile t, 16-column slice m, then the
// 64 lanes' 16-byte operand fragments (lane (r, h): columns 16 m + 8 h .. + 8 of row 32 t + r) -- so that one global_load_dwordx4 of
// a wavefront reads ONE contiguous KiB (8 cache lines) instead of 32 bytes from each of 32 rows (32 lines): with the row-major copy
// the two bf16 passes kept the texture-address unit stalled on the L1 half of the time and waited 2,000 cycles for a tile they had
// requested a whole iteration earlier (round 6: TA_ADDR_STALLED_BY_TC, SQ_WAIT_ANY; the matrix pipe at 22 / 35 %)
template <int LPR, bool TILED>
__global__ __launch_bounds__(256) void to_bf16_kernel(const float *__restrict__ X, const int32_t *__restrict__ row_ids, int rows, int rows_pad,
                                                      __bf16 *__restrict__ out, float *__restrict__ norm, float *__restrict__ norm_max) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (k < rows) x = *reinterpret_cast<const f32x4 *>(X + (int64_t)(row_ids ? row_ids[k] : k) * (4 * LPR) + 4 * r);
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 o = {(__bf16)x.x, (__bf16)x.y, (__bf16)x.z, (__bf16)x.w};
    if (k < rows_pad) {
        if constexpr (TILED) {
            constexpr int NM = LPR / 4;
            const int c = 4 * r;
            *reinterpret_cast<bf16x4 *>(out + ((((k >> 5) * NM + (c >> 4)) * 64 + ((c >> 3) & 1) * 32 + (k & 31)) << 3) + (c & 7)) = o;
        } else {
            *reinterpret_cast<bf16x4 *>(out + k * (4 * LPR) + 4 * r) = o;
        }
    }
    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    ss = row_allreduce_sum<LPR>(ss);
    float nrm = sqrtf(ss) * 1.0001f;                               // rows past `rows`: 0
    if (r == 0 && norm && k < rows_pad) norm[k] = nrm;
    if (norm_max) {                                                // one atomic per block (one per row serialises: 110 us for 38,048 rows)
        __shared__ float s_max[4];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) nrm = fmaxf(nrm, __shfl_xor(nrm, m, kWave));
        if (lane == 0) s_max[threadIdx.x >> 6] = nrm;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<int *>(norm_max), __float_as_int(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]))));   // non-negative floats order as ints
    }
}

__global__ __launch_bounds__(256) void lowered_tau_kernel(const float *__restrict__ tau, const float *__restrict__ u_norm,
                                                          const float *__restrict__ v_norm_max, int n_b, float *__restrict__ tau_low) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_b) tau_low[b] = tau[b] - 0x1p-7f * 1.02f * u_norm[b] * *v_norm_max;
}

// the same when tau_hat itself comes from bf16 scores (sample_max_bf16_kernel): N + 1 distinct unrated items have bf16 scores
// >= tau_hat, so fp32 scores >= tau_hat - eps =: tau (a threshold the user's N + 1 best fp32 scores reach); every item whose
// fp32 score reaches tau has a bf16 score >= tau - eps =: tau_low
__global__ __launch_bounds__(256) void lowered_tau2_kernel(float *__restrict__ tau, const float *__restrict__ u_norm,
                                                           const float *__restrict__ v_norm_max, int n_b, float *__restrict__ tau_low) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_b) {
        const float eps = 0x1p-7f * 1.02f * u_norm[b] * *v_norm_max, t = tau[b] - eps;
        tau[b] = t;
        tau_low[b] = t - eps;
    }
}

// sample_max_kernel_f32 on the bf16 copies, every `stride`-th item tile (stride 1: the whole catalogue -- the pass costs 1/16 of
// the fp32 matrix-pipe time, so a denser sample, a higher threshold and fewer candidates are affordable).  The accumulator
// register number rides in the low 4 mantissa bits of the score (relative 2^-19: inside eps's 2 % slack), so the per-tile
// maximum is 16 v_and_or + 8 v_max3 and the best item is recovered from (tile, register, half).
template <int NM, int NU>      // NU user tiles of 32 per wavefront
__global__ __launch_bounds__(256, 2) void sample_max_bf16_kernel(
    const __bf16 *__restrict__ Ub, const __bf16 *__restrict__ Vb, int n_items, int n_b, int b_pad, int stride, int n_s_tiles,
    int tiles_per_group, int n_groups, float *__restrict__ gmax, int32_t *__restrict__ garg) {
    constexpr int LD = 16 * NM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int g_step = gridDim.y * 4;
    int g = blockIdx.y * 4 + wave;
    if (g >= n_groups) return;
    int bs[NU];
    bf16x8 uu[NU][NM];
#pragma unroll
    for (int k = 0; k < NU; k++) {
        bs[k] = (blockIdx.x * NU + k) * 32 + r;
        const int bl = bs[k] < b_pad ? bs[k] : b_pad - 1;         // Ub holds b_pad rows (zero rows past n_b)
#pragma unroll
        for (int m = 0; m < NM; m++) uu[k][m] = *reinterpret_cast<const bf16x8 *>(Ub + (int64_t)bl * LD + 16 * m + 8 * h);
    }
    auto load_tile = [&](int s_tile, bf16x8 (&dst)[NM]) {         // Vb: whole 32-item tiles in fragment order (to_bf16_kernel<., true>)
        const __bf16 *frag = Vb + (((int64_t)s_tile * stride * NM) * 64 + lane) * 8;
#pragma unroll
        for (int m = 0; m < NM; m++) dst[m] = *reinterpret_cast<const bf16x8 *>(frag + m * 512);
    };
    auto tagged_max = [](const f32x16 &a) {
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; q++) t[q] = __uint_as_float((__float_as_uint(a[q]) & ~15u) | (unsigned)q);
        const float x0 = fmaxf(fmaxf(t[0], t[1]), t[2]), x1 = fmaxf(fmaxf(t[3], t[4]), t[5]), x2 = fmaxf(fmaxf(t[6], t[7]), t[8]);
        const float x3 = fmaxf(fmaxf(t[9], t[10]), t[11]), x4 = fmaxf(fmaxf(t[12], t[13]), t[14]);
        return fmaxf(fmaxf(fmaxf(x0, x1), fmaxf(x2, x3)), fmaxf(x4, t[15]));
    };
    bf16x8 v[NM], vn[NM];
    bool first = true;
    for (; g < n_groups; g += g_step) {
        const int s_begin = g * tiles_per_group;
        int s_end = s_begin + tiles_per_group;
        if (s_end > n_s_tiles) s_end = n_s_tiles;
        float mx[NU];
        int ts[NU];
#pragma unroll
        for (int k = 0; k < NU; k++) { mx[k] = -__builtin_huge_valf(); ts[k] = s_begin; }
        if (first) { load_tile(s_begin, v); first = false; }
        for (int st = s_begin; st < s_end; st++) {
            const int nxt = st + 1 < s_end ? st + 1 : (g + g_step) * tiles_per_group;
            const bool more = nxt < n_s_tiles && (st + 1 < s_end || g + g_step < n_groups);
            load_tile(more ? nxt : st, vn);          // unconditional: see score_filter_bf16_kernel
            f32x16 acc[NU];
#pragma unroll
            for (int k = 0; k < NU; k++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[k][q] = 0.f;
#pragma unroll
            for (int m = 0; m < NM; m++)
#pragma unroll
                for (int k = 0; k < NU; k++) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[m], uu[k][m], acc[k], 0, 0, 0);
            const int item_base = st * stride * 32 + 4 * h;
            if (st * stride * 32 + 32 > n_items) {                 // the last, partial tile: pad rows are not items
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (item_base + (q & 3) + 8 * (q >> 2) >= n_items) {
#pragma unroll
                        for (int k = 0; k < NU; k++) acc[k][q] = -__builtin_huge_valf();
                    }
            }
#pragma unroll
            for (int k = 0; k < NU; k++) {
                const float x = tagged_max(acc[k]);
                if (x > mx[k]) { mx[k] = x; ts[k] = st; }
            }
#pragma unroll
            for (int m = 0; m < NM; m++) v[m] = vn[m];
        }
#pragma unroll
        for (int k = 0; k < NU; k++) {
            const int q = (int)(__float_as_uint(mx[k]) & 15u);
            int a = ts[k] * stride * 32 + 4 * h + (q & 3) + 8 * (q >> 2);
            float m = mx[k];
            const float o = __shfl_xor(m, 32, kWave);
            const int oa = __shfl_xor(a, 32, kWave);
            if (o > m) { m = o; a = oa; }
            if (h == 0 && bs[k] < n_b) { gmax[(int64_t)g * b_pad + bs[k]] = m; garg[(int64_t)g * b_pad + bs[k]] = a; }
        }
    }
}

// score_filter2_kernel_f32's loop on the bf16 copies: NM = ld / 16 MFMAs per 32 x 32 tile (lane (r, h): columns 16 m + 8 h .. + 8 of
// its row); what reaches the LOWERED threshold goes to the lane-private lists as an id (its approximate score only for inspection)
template <int NM, int NU>      // NU user tiles of 32 per wavefront: one fetched item tile feeds NU * NM MFMAs
__global__ __launch_bounds__(256, 2) void score_filter_bf16_kernel(
    const __bf16 *__restrict__ Ub, int b_pad, const __bf16 *__restrict__ Vb, int n_items, int n_b, const float *__restrict__ tau_low, int n_lists,
    int list_cap, float *__restrict__ cand_s, int32_t *__restrict__ cand_i, int32_t *__restrict__ cand_n) {
    constexpr int LD = 16 * NM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int list = (blockIdx.y * 4 + wave) * 2 + h;
    const int n_item_tiles = (n_items + 31) / 32;
    const int t_step = gridDim.y * 4;
    const int t_begin = blockIdx.y * 4 + wave, t_end = n_item_tiles;
    int bs[NU], cnt[NU];
    bool live[NU];
#pragma unroll
    for (int k = 0; k < NU; k++) { bs[k] = (blockIdx.x * NU + k) * 32 + r; live[k] = bs[k] < n_b; cnt[k] = 0; }
    if (t_begin < t_end) {
        float th[NU];
        bf16x8 uu[NU][NM];                           // Ub holds b_pad rows (zero rows past n_b)
        float *cs[NU];
        int32_t *ci[NU];
#pragma unroll
        for (int k = 0; k < NU; k++) {
            th[k] = live[k] ? tau_low[bs[k]] : __builtin_huge_valf();
            const int bl = bs[k] < b_pad ? bs[k] : b_pad - 1;
#pragma unroll
            for (int m = 0; m < NM; m++) uu[k][m] = *reinterpret_cast<const bf16x8 *>(Ub + (int64_t)bl * LD + 16 * m + 8 * h);
            cs[k] = cand_s + ((int64_t)bs[k] * n_lists + list) * list_cap;
            ci[k] = cand_i + ((int64_t)bs[k] * n_lists + list) * list_cap;
        }
        auto load_tile = [&](int t, bf16x8 (&dst)[NM]) {          // Vb: whole 32-item tiles (zero pad rows) in fragment order
            const __bf16 *frag = Vb + (((int64_t)t * NM) * 64 + lane) * 8;
#pragma unroll
            for (int m = 0; m < NM; m++) dst[m] = *reinterpret_cast<const bf16x8 *>(frag + m * 512);
        };
        auto do_tile = [&](int t, const bf16x8 (&v)[NM]) {
            f32x16 acc[NU];
#pragma unroll
            for (int k = 0; k < NU; k++)
#pragma unroll
                for (int q = 0; q < 16; q++) acc[k][q] = 0.f;
#pragma unroll
            for (int m = 0; m < NM; m++)
#pragma unroll
                for (int k = 0; k < NU; k++) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[m], uu[k][m], acc[k], 0, 0, 0);
            const int item_base = t * 32 + 4 * h;
            if (t * 32 + 32 > n_items) {              // the last, partial tile: its pad rows are never candidates
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (item_base + (q & 3) + 8 * (q >> 2) >= n_items) {
#pragma unroll
                        for (int k = 0; k < NU; k++) acc[k][q] = -__builtin_huge_valf();
                    }
            }
            // hits are rare (about one per 32 x 32 tile): a tile is tested by the maximum of its 16 accumulator registers (v_max3
            // chains, one compare, one scalar branch on the wave's mask); a tile with a hit narrows down by register quads --
            // not 16 dependent compare / branch pairs per tile, which cost 2.5x the tile's MFMA time
#pragma unroll
            for (int k = 0; k < NU; k++) {
                const f32x16 &a = acc[k];
                float gq[4];
#pragma unroll
                for (int j = 0; j < 4; j++) gq[j] = fmaxf(fmaxf(fmaxf(a[4 * j], a[4 * j + 1]), a[4 * j + 2]), a[4 * j + 3]);
                const float mx = fmaxf(fmaxf(fmaxf(gq[0], gq[1]), gq[2]), gq[3]);
                if (__ballot(mx >= th[k])) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (__ballot(gq[j] >= th[k])) {
#pragma unroll
                            for (int q = 4 * j; q < 4 * j + 4; q++)
                                if (a[q] >= th[k]) {
                                    if (cnt[k] < list_cap) { cs[k][cnt[k]] = a[q]; ci[k][cnt[k]] = item_base + (q & 3) + 8 * (q >> 2); }
                                    cnt[k]++;
                                }
                        }
                }
            }
        };
        // The next tile is fetched UNCONDITIONALLY (past the end: the last tile again, unused).  Under `if (t + t_step < t_end)` the
        // compiler cannot know at the merge how many loads are outstanding, assumes the fewest, and waits for the tile it has just
        // requested together with the one it is about to use (s_waitcnt vmcnt(3..0) in front of the MFMAs): no software pipelining
        // (round 6; the same finding as the BPR kernels' 64-bit flavour).  All four tile loops of this file fetch this way.
        bf16x8 va[NM], vb[NM];
        load_tile(t_begin, va);
        int t = t_begin;
        while (true) {
            load_tile(t + t_step < t_end ? t + t_step : t_end - 1, vb);
            do_tile(t, va);
            t += t_step;
            if (t >= t_end) break;
            load_tile(t + t_step < t_end ? t + t_step : t_end - 1, va);
            do_tile(t, vb);
            t += t_step;
            if (t >= t_end) break;
        }
    }
#pragma unroll
    for (int k = 0; k < NU; k++)
        if (live[k]) cand_n[(int64_t)bs[k] * n_lists + list] = cnt[k];
}

constexpr int kSelectWaves = 4;
template <int NC>       // column chunks of 64 in the re-scoring (ld <= 64: 1, ld = 128: 2)
__global__ __launch_bounds__(64 * kSelectWaves) void select_topk_kernel(
    const float *__restrict__ cand_s, const int32_t *__restrict__ cand_i, const int32_t *__restrict__ cand_n, int n_lists,
    const float *__restrict__ tau, const int32_t *__restrict__ user_ids, const int64_t *__restrict__ rated_indptr,
    const int32_t *__restrict__ rated_sorted, int n_b, int K, int32_t *__restrict__ ids_out, float *__restrict__ scores_out,
    int32_t *__restrict__ flags, int32_t *__restrict__ n_flagged, int32_t *__restrict__ flagged_list,
    const float *__restrict__ U, const float *__restrict__ V, int ld, int pool_cap, const float *__restrict__ tau_low, int list_cap) {
    // pool_cap: LDS entries per user (a user with more candidates is flagged and redone exactly); the lists' worst case,
    // n_lists * list_cap, would leave one wavefront per SIMD.
    // U != null: the candidates come from the bf16 filter -- their fp32 scores are formed here, by the MFMA sequence of
    // score_kernel_f32 (32 candidates as the item rows of a tile, the user's row broadcast over its 32 columns)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * kSelectWaves + wave;
    if (b >= n_b) return;                                         // whole wavefront
    unsigned long long *pool = reinterpret_cast<unsigned long long *>(smem) + (int64_t)wave * pool_cap;
    // the re-scoring's user operand (the same for every tile): fetched first, its latency under the pooling below
    const int r = lane & 31, h = lane >> 5;
    f32x4 uu[NC][8];
    int kbs[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int col0 = 64 * c + 32 * h;
        const bool kv = col0 < ld;
        kbs[c] = kv ? col0 : 0;
        const float keep = kv ? 1.f : 0.f;
        const f32x4 *pu = reinterpret_cast<const f32x4 *>((U ? U : V) + (U ? (int64_t)user_ids[b] : 0) * ld + kbs[c]);
#pragma unroll
        for (int q = 0; q < 8; q++) uu[c][q] = pu[q] * keep;
    }
    // key: score mapped to an order-preserving unsigned, then ~id so that among equal scores the LOWER id is larger
    auto key_of = [](float s, int32_t id) {
        unsigned u = __float_as_uint(s);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        return ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - id);
    };
    bool bad = !(tau[b] > 0.f);
    int64_t rbeg = 0, rend = 0;
    if (rated_indptr) { const int uid = user_ids[b]; rbeg = rated_indptr[uid]; rend = rated_indptr[uid + 1]; }
    int total = 0, kept = 0;
    for (int l0 = 0; l0 < n_lists; l0 += 64) {                    // counts -> offsets (wave scan), candidates -> pool
        const int l = l0 + lane;
        int n = l < n_lists ? cand_n[(int64_t)b * n_lists + l] : 0;
        if (n > list_cap) { bad = true; n = list_cap; }
        int inc = n;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(inc, sft, 64);
            if (lane >= sft) inc += o;
        }
        const int off = total + inc - n;
        for (int c = 0; c < n; c++) {
            const int64_t at = ((int64_t)b * n_lists + l) * list_cap + c;
            if (off + c < pool_cap) pool[off + c] = key_of(cand_s[at], cand_i[at]);
        }
        total += __shfl(inc, 63, 64);
    }
    if (total > pool_cap) { bad = true; total = pool_cap; }
    // rated items are masked to 0 < tau: not candidates.  The lists are uneven, the pool is not: every lane bisects its
    // share of the pool (a few elements each)
    for (int e = lane; e < total; e += 64) {
        const int32_t item = 0x7fffffff - (int32_t)(unsigned)(pool[e] & 0xffffffffu);
        int64_t lo = rbeg, hi = rend;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rated_sorted[mid] < item) lo = mid + 1; else hi = mid;
        }
        const bool rated = lo < rend && rated_sorted[lo] == item;
        if (rated) pool[e] = 0ull;
        kept += rated ? 0 : 1;
    }
    if (U) {
        // Which candidates can still be among the N + 1 best?  Their bf16 scores s^ are within eps of the fp32 scores.  Let L be
        // the (N+1)-th largest s^: N + 1 candidates have fp32 scores >= L - eps, so the fp32 (N+1)-th best is >= L - eps and a
        // candidate with s^ < L - 2 eps (fp32 score < L - eps) is out.  Only the others are re-scored: ~30 of ~250.
        const float eps = (tau[b] - tau_low[b]) * 1.01f;
        const int Mq = K + 1;
        unsigned long long below = ~0ull;
        if (total <= 64) {                                         // the usual case: one key per lane, ranks by v_readlane
            const unsigned long long key = lane < total ? pool[lane] : 0ull;
            const int rk = wave_rank_u64(key);
            const unsigned long long at = __ballot(key != 0ull && rk == Mq - 1);
            below = at ? wave_read_u64(key, __builtin_ctzll(at)) : 0ull;
        } else
        for (int rank = 0; rank < Mq; rank++) {                    // the Mq-th largest key, non-destructively
            unsigned long long best = 0;
            for (int e = lane; e < total; e += 64) {
                const unsigned long long k = pool[e];
                if (k < below && k > best) best = k;
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const unsigned long long o = __shfl_xor(best, m, 64);
                best = o > best ? o : best;
            }
            below = best;
            if (!best) break;                                      // fewer than Mq candidates (only after an overflow)
        }
        if (below) {
            unsigned ub_ = (unsigned)(below >> 32);
            ub_ = (ub_ & 0x80000000u) ? (ub_ & 0x7fffffffu) : ~ub_;
            const float cut = __uint_as_float(ub_) - 2.f * eps;
            const unsigned long long cut_key = key_of(cut, 0x7fffffff);          // the smallest key with that score
            int n_keep = 0;
            for (int e0 = 0; e0 < total; e0 += 64) {               // in-place forward compaction, chunk by chunk
                const int e = e0 + lane;
                const unsigned long long k = e < total ? pool[e] : 0ull;
                const bool keep = k != 0ull && k >= cut_key;
                const unsigned long long mask = __ballot(keep);
                const int pos = n_keep + __popcll(mask & ((1ull << lane) - 1ull));
                if (keep) pool[pos] = k;                           // pos <= e: never ahead of what is still to be read
                n_keep += __popcll(mask);
            }
            total = n_keep;
            kept = lane == 0 ? n_keep : 0;                         // survivors are unrated: the count the checks below sum up
        }
        auto item_of = [&](int e) -> int64_t {
            const unsigned long long k = e < total ? pool[e] : 0ull;
            return k ? (int64_t)(0x7fffffff - (int32_t)(unsigned)(k & 0xffffffffu)) : 0;
        };
        auto load_rows = [&](int64_t item, f32x4 (&dst)[NC][8]) {
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const f32x4 *pv = reinterpret_cast<const f32x4 *>(V + item * ld + kbs[c]);
#pragma unroll
                for (int q = 0; q < 8; q++) dst[c][q] = pv[q];
            }
        };
        f32x4 vv[NC][8], vn[NC][8];
        load_rows(item_of(r), vv);
        for (int e0 = 0; e0 < total; e0 += 32) {
            if (e0 + 32 < total) load_rows(item_of(e0 + 32 + r), vn);        // the next tile's rows under this tile's MFMAs
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++)                                // the operand walk of score_kernel_f32, term for term
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[c][q].x, uu[c][q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[c][q].y, uu[c][q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[c][q].z, uu[c][q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[c][q].w, uu[c][q].w, acc, 0, 0, 0);
                }
            // every column of the tile is this user: lanes 0 and 32 hold the 32 rows between them
            if (r == 0) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int ee = e0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (ee < total) {
                        const unsigned long long old = pool[ee];
                        if (old) pool[ee] = key_of(acc[q], 0x7fffffff - (int32_t)(unsigned)(old & 0xffffffffu));
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int q = 0; q < 8; q++) vv[c][q] = vn[c][q];
        }
    }
    bad = __any(bad);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) kept += __shfl_xor(kept, m, 64);
    const int M = K + 1;
    if (kept < M) bad = true;                                     // only after an overflow
    // each lane owns pool[lane], pool[lane + 64], ...
    unsigned long long prev = 0;
    bool tie = false;
    const int take = kept < M ? kept : M;
    if (total <= 64) {                                            // one key per lane: its rank is its place in the output
        const unsigned long long key = lane < total ? pool[lane] : 0ull;
        const int rk = wave_rank_u64(key);                        // rated / empty slots (0) rank behind every candidate
        if (key != 0ull && rk < take) pool[rk] = key;             // every lane has read its key: the pool can take the sorted run
        if (key != 0ull && rk < K && rk < take) {
            unsigned u = (unsigned)(key >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            ids_out[(int64_t)b * K + rk] = 0x7fffffff - (int32_t)(unsigned)(key & 0xffffffffu);
            scores_out[(int64_t)b * K + rk] = __uint_as_float(u);
        }
        tie = __any(lane >= 1 && lane < take && (pool[lane] >> 32) == (pool[lane - 1] >> 32));
    } else
    for (int rank = 0; rank < take; rank++) {
        unsigned long long best = 0;
        int where = -1;
        for (int e = lane; e < total; e += 64) {
            const unsigned long long k = pool[e];
            if (k > best) { best = k; where = e; }
        }
        unsigned long long wbest = best;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = __shfl_xor(wbest, m, 64);
            wbest = o > wbest ? o : wbest;
        }
        if (best == wbest && where >= 0) pool[where] = 0;          // keys are unique (ids differ): exactly one lane
        if (rank > 0 && (wbest >> 32) == (prev >> 32)) tie = true;
        prev = wbest;
        if (rank < K && lane == 0) {
            unsigned u = (unsigned)(wbest >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            ids_out[(int64_t)b * K + rank] = 0x7fffffff - (int32_t)(unsigned)(wbest & 0xffffffffu);
            scores_out[(int64_t)b * K + rank] = __uint_as_float(u);
        }
    }
    if (lane == 0) {
        const bool flag = bad || tie;
        flags[b] = flag ? 1 : 0;
        if (flag) flagged_list[atomicAdd(n_flagged, 1)] = b;
    }
}

__global__ void gather_user_ids_kernel(const int32_t *__restrict__ user_ids, const int32_t *__restrict__ flagged_list, int off, int n,
                                       int32_t *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = user_ids[flagged_list[off + k]];
}
__global__ void scatter_rows_kernel(const int32_t *__restrict__ flagged_list, int off, int n, int K, const int32_t *__restrict__ ids_in,
                                    const float *__restrict__ sc_in, int32_t *__restrict__ ids_out, float *__restrict__ sc_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * K) return;
    const int64_t dst = (int64_t)flagged_list[off + t / K] * K + t % K;
    ids_out[dst] = ids_in[t];
    sc_out[dst] = sc_in[t];
}

template <typename T>
int run_score_topk(const void *U, const void *V, int d, int ld, int n_items, const int32_t *user_ids,
                   int n_b, const int64_t *rated_indptr, const int32_t *rated_items, int K, void *scratch,
                   int32_t *ids_out, void *scores_out, hipStream_t st) {
    constexpr int TILE = sizeof(T) == 4 ? 32 : 16;         // items per MFMA tile
    constexpr int UTILE = sizeof(T) == 4 ? 64 : 16;        // users per wavefront (fp32: two 32-user tiles)
    const int b_pad = (n_b + 63) / 64 * 64;
    const int n_utiles = (n_b + UTILE - 1) / UTILE;
    const int n_item_tiles = (n_items + TILE - 1) / TILE;
    // enough waves to fill 256 CUs x 4 SIMDs a few times over, each streaming >= 8 item tiles
    int splits = (4096 + n_utiles - 1) / n_utiles;          // item-range splits per user tile, in waves
    int per_wave = (n_item_tiles + splits - 1) / splits;
    if (per_wave < 8) per_wave = n_item_tiles < 8 ? n_item_tiles : 8;
    const int waves_per_utile = (n_item_tiles + per_wave - 1) / per_wave;
    const dim3 grid((unsigned)n_utiles, (unsigned)((waves_per_utile + 3) / 4));
    T *S_T = static_cast<T *>(scratch);
    if constexpr (sizeof(T) == 4)
        hipLaunchKernelGGL(score_kernel_f32, grid, dim3(256), 0, st, (const float *)U, (const float *)V, ld,
                           n_items, user_ids, n_b, b_pad, per_wave, S_T, 1, (int64_t)64, (int64_t)1);
    else
        hipLaunchKernelGGL(score_kernel_f64, grid, dim3(256), 0, st, (const double *)U, (const double *)V, ld,
                           n_items, user_ids, n_b, b_pad, per_wave, S_T);
    QREC_LAUNCH_CHECK();
    if (rated_indptr) {
        hipLaunchKernelGGL(mask_kernel<T>, dim3((unsigned)((n_b + 3) / 4)), dim3(256), 0, st, user_ids, n_b,
                           rated_indptr, rated_items, b_pad, S_T, TILE, 1, (int64_t)64, (int64_t)1);
        QREC_LAUNCH_CHECK();
    }
    const size_t lds = (size_t)K * kHeapThreads * (sizeof(T) + sizeof(int32_t));
    QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&heap_topk_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned user_blocks = (unsigned)((n_b + kHeapThreads - 1) / kHeapThreads);
    const int M = K + 1;
    if (n_items < 8192 || M > kGroups) {   // short catalogue, or more candidates wanted than there are group maxima:
                                           // the sequential emulation for everybody
        hipLaunchKernelGGL(heap_topk_kernel<T>, dim3(user_blocks), dim3(kHeapThreads), lds, st, S_T, n_items, b_pad, n_b, K,
                           ids_out, (T *)scores_out, nullptr, nullptr, 0);
        QREC_LAUNCH_CHECK();
        return QREC_OK;
    }
    const int per_group = (n_items + kGroups - 1) / kGroups, n_groups_used = (n_items + per_group - 1) / per_group;
    const int per_slice = (n_items + kSlices - 1) / kSlices;
    // scratch behind the score block: group maxima [kGroups][b_pad], tau, candidate rows [kSlices*kSliceCap][b_pad] (scores,
    // ids), per-slice counts, flags, flagged list, counter
    unsigned char *extra = static_cast<unsigned char *>(scratch) + score_block_bytes(sizeof(T), n_items, n_b);
    const size_t cand_rows = (size_t)kSlices * kSliceCap;
    T *gmax = reinterpret_cast<T *>(extra);
    T *tau = gmax + (size_t)kGroups * b_pad;
    T *cand_s = tau + b_pad;
    int32_t *cand_i = reinterpret_cast<int32_t *>(cand_s + cand_rows * b_pad);
    int32_t *cand_n = cand_i + cand_rows * b_pad;
    int32_t *flags = cand_n + (size_t)kSlices * b_pad;
    int32_t *flagged_list = flags + b_pad;
    int32_t *n_flagged = flagged_list + b_pad;
    QREC_HIP_CHECK(hipMemsetAsync(n_flagged, 0, sizeof(int32_t), st));
    const size_t lds_m = (size_t)M * kHeapThreads * (sizeof(T) + sizeof(int32_t));
    QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_topk_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    const unsigned lane_blocks = (unsigned)((n_b + 255) / 256);
    if (n_groups_used < kGroups)      // ranges past the end (n_items not a multiple): -inf maxima
        hipLaunchKernelGGL(fill_neg_inf_kernel<T>, dim3(lane_blocks, kGroups - n_groups_used), dim3(256), 0, st,
                           gmax + (size_t)n_groups_used * b_pad, b_pad, n_b);
    hipLaunchKernelGGL(group_max_kernel<T>, dim3(lane_blocks, (unsigned)n_groups_used), dim3(256), 0, st, S_T, n_items, b_pad, n_b,
                       per_group, gmax);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(threshold_kernel<T>, dim3(lane_blocks), dim3(256), 0, st, gmax, b_pad, n_b, M, tau);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(filter_kernel<T>, dim3(lane_blocks, kSlices), dim3(256), 0, st, S_T, n_items, b_pad, n_b, per_slice, tau,
                       cand_s, cand_i, cand_n);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(merge_topk_kernel<T>, dim3(user_blocks), dim3(kHeapThreads), lds_m, st, cand_s, cand_i, cand_n, b_pad, n_b, K,
                       ids_out, (T *)scores_out, flags, n_flagged, flagged_list);
    QREC_LAUNCH_CHECK();
    // users whose N+1 best scores are not pairwise distinct: exact emulation (device-side choice of the regime)
    const int many = n_b / 16 > 256 ? n_b / 16 : 256;
    const int wave_blocks = n_b < many ? n_b : many;
    if (K <= 64)
        hipLaunchKernelGGL((exact_wave_kernel<T, true>), dim3((unsigned)wave_blocks), dim3(64), 0, st, S_T,
                           n_items, b_pad, K, n_flagged, flagged_list, many, ids_out, (T *)scores_out, (int64_t)64, (int64_t)1);
    else
        hipLaunchKernelGGL((exact_wave_kernel<T, false>), dim3((unsigned)wave_blocks), dim3(64), (size_t)K * (sizeof(T) + sizeof(int32_t)), st, S_T,
                           n_items, b_pad, K, n_flagged, flagged_list, many, ids_out, (T *)scores_out, (int64_t)64, (int64_t)1);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(heap_topk_kernel<T>, dim3(user_blocks), dim3(kHeapThreads), lds, st, S_T, n_items, b_pad, n_b, K, ids_out,
                       (T *)scores_out, flags, n_flagged, many);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

// geometry shared by the scratch-size query and the launch
// ---- (A) of the fused evaluation: the threshold without a sampled score block --------------------------------------------
// sample_max_kernel_f32: the scoring loop over every kSampleStride-th item tile, but nothing is stored: per GROUP of
// `tiles_per_group` consecutive sampled tiles every user keeps the largest (unmasked) score and its item.
// threshold_var_kernel: a group whose best item is one the user has rated is discarded -- the remaining maxima are
// scores of DISTINCT UNRATED items, so their (N+1)-th largest is a threshold the user's N+1 best masked scores reach.
// (First version: the sampled tiles written out as a 600 MB block, masked there, group maxima in a second streaming pass:
// 0.45 ms of the evaluation; a discarded group only lowers tau a little: ~7 % of the groups of a 40-item user.)
constexpr int kMaxGroups = 128;
template <int NC>
__global__ __launch_bounds__(256) void sample_max_kernel_f32(
    const float *__restrict__ U, const float *__restrict__ V, int ld, int n_items, const int32_t *__restrict__ user_ids,
    int n_b, int b_pad, int n_s_tiles, int tiles_per_group, int n_groups, float *__restrict__ gmax, int32_t *__restrict__ garg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int upair = blockIdx.x;
    const int b0 = upair * 64 + r, b1 = b0 + 32;
    const bool live0 = b0 < n_b, live1 = b1 < n_b;
    const int64_t uid0 = user_ids[live0 ? b0 : n_b - 1], uid1 = user_ids[live1 ? b1 : n_b - 1];
    const int g_step = gridDim.y * 4;
    int g = blockIdx.y * 4 + wave;
    if (g >= n_groups) return;
    f32x4 ua[NC][8], ub[NC][8];
    int kb[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int col0 = 64 * c + 32 * h;
        const bool kv = col0 < ld;
        kb[c] = kv ? col0 : 0;
        const float keep = kv ? 1.f : 0.f;
        const f32x4 *p0 = reinterpret_cast<const f32x4 *>(U + uid0 * ld + kb[c]), *p1 = reinterpret_cast<const f32x4 *>(U + uid1 * ld + kb[c]);
#pragma unroll
        for (int q = 0; q < 8; q++) { ua[c][q] = p0[q] * keep; ub[c][q] = p1[q] * keep; }
    }
    auto load_tile = [&](int s_tile, f32x4 (&dst)[NC][8]) {        // sampled tile s -> item tile s * kSampleStride
        const int item = s_tile * kSampleStride * 32 + r;
        const float *row = V + (int64_t)(item < n_items ? item : n_items - 1) * ld;
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int q = 0; q < 8; q++) dst[c][q] = reinterpret_cast<const f32x4 *>(row + kb[c])[q];
    };
    f32x4 v[NC][8], vn[NC][8];
    bool first = true;
    for (; g < n_groups; g += g_step) {
        const int s_begin = g * tiles_per_group;
        int s_end = s_begin + tiles_per_group;
        if (s_end > n_s_tiles) s_end = n_s_tiles;
        float m0 = -__builtin_huge_valf(), m1 = m0;
        int a0 = 0, a1 = 0;
        if (first) { load_tile(s_begin, v); first = false; }
        for (int st = s_begin; st < s_end; st++) {
            // the next tile to come (this group's, or the first of the wavefront's next group) loads under this tile's MFMAs
            const int nxt = st + 1 < s_end ? st + 1 : (g + g_step) * tiles_per_group;
            const bool more = nxt < n_s_tiles && (st + 1 < s_end || g + g_step < n_groups);
            load_tile(more ? nxt : st, vn);          // unconditional: see score_filter_bf16_kernel
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; q++) { acc0[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].x, ua[c][q].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].x, ub[c][q].x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].y, ua[c][q].y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].y, ub[c][q].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].z, ua[c][q].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].z, ub[c][q].z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].w, ua[c][q].w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c][q].w, ub[c][q].w, acc1, 0, 0, 0);
                }
            const int item_base = st * kSampleStride * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int item = item_base + (q & 3) + 8 * (q >> 2);
                const bool in = item < n_items;                    // rows past the catalogue repeat its last item
                if (in && acc0[q] > m0) { m0 = acc0[q]; a0 = item; }
                if (in && acc1[q] > m1) { m1 = acc1[q]; a1 = item; }
            }
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int q = 0; q < 8; q++) v[c][q] = vn[c][q];
        }
        // the two k-slot halves of a column hold different item rows of the same user: fold them
        const float o0 = __shfl_xor(m0, 32, kWave), o1 = __shfl_xor(m1, 32, kWave);
        const int oa0 = __shfl_xor(a0, 32, kWave), oa1 = __shfl_xor(a1, 32, kWave);
        if (o0 > m0) { m0 = o0; a0 = oa0; }
        if (o1 > m1) { m1 = o1; a1 = oa1; }
        if (h == 0) {
            if (live0) { gmax[(int64_t)g * b_pad + b0] = m0; garg[(int64_t)g * b_pad + b0] = a0; }
            if (live1) { gmax[(int64_t)g * b_pad + b1] = m1; garg[(int64_t)g * b_pad + b1] = a1; }
        }
    }
}

// tau[b] = the M-th largest of the user's n_groups (<= kMaxGroups) group maxima, not counting groups whose best item is one the
// user has rated (bisection in the user's sorted rated items -- only for the groups that come up, ~M of them: as a kernel of its
// own over all (group, user) pairs the check was 88 us)
__global__ __launch_bounds__(256) void threshold_var_kernel(const float *__restrict__ gmax, const int32_t *__restrict__ garg,
                                                            const int32_t *__restrict__ user_ids, const int64_t *__restrict__ rated_indptr,
                                                            const int32_t *__restrict__ rated_sorted, int b_pad, int n_b, int n_groups, int M,
                                                            float *__restrict__ tau) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_b) return;
    float v[kMaxGroups];
#pragma unroll
    for (int g = 0; g < kMaxGroups; g++) v[g] = g < n_groups ? gmax[(int64_t)g * b_pad + b] : -__builtin_huge_valf();
    int64_t rbeg = 0, rend = 0;
    if (rated_indptr) { const int uid = user_ids[b]; rbeg = rated_indptr[uid]; rend = rated_indptr[uid + 1]; }
    float th = -__builtin_huge_valf();
    int found = 0;
    for (int it = 0; it < n_groups && found < M; it++) {
        int arg = 0;
        float best = v[0];
#pragma unroll
        for (int g = 1; g < kMaxGroups; g++)
            if (v[g] > best) { best = v[g]; arg = g; }
#pragma unroll
        for (int g = 0; g < kMaxGroups; g++)
            if (g == arg) v[g] = -__builtin_huge_valf();
        if (!(best > -__builtin_huge_valf())) break;
        if (rated_indptr) {
            const int item = garg[(int64_t)arg * b_pad + b];
            int64_t lo = rbeg, hi = rend;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (rated_sorted[mid] < item) lo = mid + 1; else hi = mid;
            }
            if (lo < rend && rated_sorted[lo] == item) continue;   // the group's best is a rated item: the group is out
        }
        th = best;
        found++;
    }
    tau[b] = found == M ? th : -__builtin_huge_valf();            // fewer than M groups left: tau <= 0, the user goes the exact way
}

struct FusedGeom {
    int b_pad, n_utiles, n_item_tiles, grid_x, grid_y, n_lists, list_cap, nu, n_s_tiles, n_s, fb_users;
    bool use_bf16;
    size_t off_ub, off_vb, off_un, off_vmax, off_taul, off_gmax, off_garg, off_tau, off_cs, off_ci, off_cn, off_flags, off_list, off_nf, off_fbu, off_fbi, off_fbs, off_fb, total;
};
__host__ inline bool fused_ok(int dtype, int ld, int n_items, int K) {
    // one predicate for the scratch-size query and the launch (QREC_EVAL_BLOCK_PATH forces the block route in both)
    return dtype == QREC_F32 && ld <= 128 && K + 1 <= kGroups && n_items >= 64 * kSampleStride * 32 && !getenv("QREC_EVAL_BLOCK_PATH");
}
__host__ inline size_t block_path_bytes(size_t elem, int n_items, int n_b) {
    const size_t b_pad = ((size_t)n_b + 63) / 64 * 64;
    return score_block_bytes(elem, n_items, n_b) + (size_t)(kGroups + 1) * b_pad * elem + (size_t)kSlices * kSliceCap * b_pad * (elem + 4) +
           (size_t)(kSlices + 2) * b_pad * 4 + 64;
}
__host__ inline FusedGeom fused_geometry(int n_items, int n_b, int ld) {
    FusedGeom g;
    g.b_pad = (n_b + 63) / 64 * 64;
    g.n_utiles = (n_b + 63) / 64;
    g.n_item_tiles = (n_items + 31) / 32;
    // the bf16 route (default): NU user tiles of 32 per wavefront, short candidate lists (its threshold comes from the whole
    // catalogue: a few dozen candidates per user); the fp32 route (QREC_EVAL_F32_FILTER): two tiles, ~200 candidates per user
    const bool bf16_filter = getenv("QREC_EVAL_F32_FILTER") == nullptr;          // read per call: the tests switch routes in-process
    const int eval_nu = getenv("QREC_EVAL_NU") ? atoi(getenv("QREC_EVAL_NU")) : 4;
    g.use_bf16 = bf16_filter && (ld == 32 || ld == 64 || ld == 128);
    g.nu = g.use_bf16 ? (ld == 128 || (ld == 64 && eval_nu == 2) ? 2 : 4) : 2;
    g.list_cap = g.use_bf16 ? 16 : kListCap;
    g.grid_x = (n_b + 32 * g.nu - 1) / (32 * g.nu);
    int waves = ((g.use_bf16 ? 8192 : 4096) + g.grid_x - 1) / g.grid_x;   // wavefronts per user group: fill 1,024 SIMDs a few times over
    if (waves > g.n_item_tiles / 8) waves = g.n_item_tiles / 8 > 0 ? g.n_item_tiles / 8 : 1;
    if (waves > 32) waves = 32;              // <= 64 candidate lists per user: the selection kernel gathers them in LDS
    g.grid_y = (waves + 3) / 4;
    g.n_lists = g.grid_y * 4 * 2;
    g.n_s_tiles = (g.n_item_tiles + kSampleStride - 1) / kSampleStride;
    const int first_of_last = (g.n_s_tiles - 1) * kSampleStride * 32;
    const int valid_last = n_items - first_of_last < 32 ? n_items - first_of_last : 32;
    g.n_s = (g.n_s_tiles - 1) * 32 + valid_last;
    g.fb_users = (n_b / 16 > 256 ? n_b / 16 : 256);
    if (g.fb_users > n_b) g.fb_users = n_b;
    g.fb_users = (g.fb_users + 63) / 64 * 64;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t o = 0;
    g.off_ub = o; o += up((size_t)g.b_pad * ld * 2);                          // bf16 copies of the batch's user rows and of the items
    g.off_vb = o; o += up((size_t)g.n_item_tiles * 32 * ld * 2);
    g.off_un = o; o += up((size_t)g.b_pad * 4);
    g.off_vmax = o; o += 256;
    g.off_taul = o; o += up((size_t)g.b_pad * 4);
    g.off_gmax = o; o += up((size_t)kMaxGroups * g.b_pad * 4);
    g.off_garg = o; o += up((size_t)kMaxGroups * g.b_pad * 4);
    g.off_tau = o; o += up((size_t)g.b_pad * 4);
    g.off_cs = o; o += up((size_t)g.b_pad * g.n_lists * g.list_cap * 4);
    g.off_ci = o; o += up((size_t)g.b_pad * g.n_lists * g.list_cap * 4);
    g.off_cn = o; o += up((size_t)g.b_pad * g.n_lists * 4);
    g.off_flags = o; o += up((size_t)g.b_pad * 4);
    g.off_list = o; o += up((size_t)g.b_pad * 4);
    g.off_nf = o; o += 256;
    g.off_fbu = o; o += up((size_t)g.fb_users * 4);
    g.off_fbi = o; o += up((size_t)g.fb_users * 100 * 4);
    g.off_fbs = o; o += up((size_t)g.fb_users * 100 * 4);
    g.off_fb = o; o += up(block_path_bytes(4, n_items, g.fb_users));
    g.total = o;
    return g;
}

int run_fused_topk_f32(const float *U, const float *V, int d, int ld, int n_items, const int32_t *user_ids, int n_b,
                       const int64_t *rated_indptr, const int32_t *rated_sorted, int K, void *scratch, int32_t *ids_out,
                       float *scores_out, hipStream_t st) {
    const FusedGeom g = fused_geometry(n_items, n_b, ld);
    unsigned char *base = static_cast<unsigned char *>(scratch);
    float *gmax = reinterpret_cast<float *>(base + g.off_gmax), *tau = reinterpret_cast<float *>(base + g.off_tau);
    float *cand_s = reinterpret_cast<float *>(base + g.off_cs);
    int32_t *cand_i = reinterpret_cast<int32_t *>(base + g.off_ci), *cand_n = reinterpret_cast<int32_t *>(base + g.off_cn);
    int32_t *flags = reinterpret_cast<int32_t *>(base + g.off_flags), *flagged_list = reinterpret_cast<int32_t *>(base + g.off_list);
    int32_t *n_flagged = reinterpret_cast<int32_t *>(base + g.off_nf);
    const int M = K + 1;
    const int bf16_stride_env = getenv("QREC_EVAL_BF16_STRIDE") ? atoi(getenv("QREC_EVAL_BF16_STRIDE")) : 1;
    const bool use_bf16 = g.use_bf16;
    const int bf16_stride = bf16_stride_env < 1 ? 1 : (bf16_stride_env > kSampleStride ? kSampleStride : bf16_stride_env);
    __bf16 *Ub = reinterpret_cast<__bf16 *>(base + g.off_ub), *Vb = reinterpret_cast<__bf16 *>(base + g.off_vb);
    float *u_norm = reinterpret_cast<float *>(base + g.off_un), *v_max = reinterpret_cast<float *>(base + g.off_vmax);
    float *tau_low = reinterpret_cast<float *>(base + g.off_taul);
    if (use_bf16) {
        // bf16 copies of the batch's user rows (gathered) and of the items, with |u_b| and max |v|
        QREC_HIP_CHECK(hipMemsetAsync(v_max, 0, sizeof(float), st));
        const int v_rows_pad = g.n_item_tiles * 32;
#define QREC_BF(LPR)                                                                                                                  \
        hipLaunchKernelGGL((to_bf16_kernel<LPR, false>), dim3((unsigned)((g.b_pad + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))), dim3(256), 0, st, U,   \
                           user_ids, n_b, g.b_pad, Ub, u_norm, (float *)nullptr);                                                     \
        hipLaunchKernelGGL((to_bf16_kernel<LPR, true>), dim3((unsigned)((v_rows_pad + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))), dim3(256), 0, st, V, \
                           (const int32_t *)nullptr, n_items, v_rows_pad, Vb, (float *)nullptr, v_max)
        if (ld == 32) { QREC_BF(8); } else if (ld == 64) { QREC_BF(16); } else { QREC_BF(32); }
#undef QREC_BF
        QREC_LAUNCH_CHECK();
    }
    // (A) threshold: group maxima straight from a scoring loop -- fp32 over every kSampleStride-th item tile, or (bf16 route)
    // bf16 over every bf16_stride-th tile (default: all of them)
    {
        const int n_s_tiles = use_bf16 ? (g.n_item_tiles + bf16_stride - 1) / bf16_stride : g.n_s_tiles;
        int tpg = use_bf16 ? (n_s_tiles + kMaxGroups - 1) / kMaxGroups : n_s_tiles / 64;
        if (tpg < 1) tpg = 1;
        int n_groups = (n_s_tiles + tpg - 1) / tpg;
        while (n_groups > kMaxGroups) { tpg++; n_groups = (n_s_tiles + tpg - 1) / tpg; }
        int waves = ((use_bf16 ? 8192 : 2048) + g.grid_x - 1) / g.grid_x;           // a few groups per wavefront: its user rows are fetched once
        if (waves > n_groups) waves = n_groups;
        if (waves < 1) waves = 1;
        const dim3 sgrid((unsigned)g.grid_x, (unsigned)((waves + 3) / 4));
        int32_t *garg = reinterpret_cast<int32_t *>(base + g.off_garg);
        if (use_bf16) {
#define QREC_SM(NM, NU)                                                                                                               \
            hipLaunchKernelGGL((sample_max_bf16_kernel<NM, NU>), sgrid, dim3(256), 0, st, Ub, Vb,                                      \
                               n_items, n_b, g.b_pad, bf16_stride, n_s_tiles, tpg, n_groups, gmax, garg)
            if (ld == 32) { QREC_SM(2, 4); }
            else if (ld == 64) { if (g.nu == 2) { QREC_SM(4, 2); } else { QREC_SM(4, 4); } }
            else { QREC_SM(8, 2); }
#undef QREC_SM
        } else if (ld <= 64)
            hipLaunchKernelGGL(sample_max_kernel_f32<1>, sgrid, dim3(256), 0, st, U, V, ld, n_items, user_ids, n_b, g.b_pad, n_s_tiles, tpg, n_groups, gmax, garg);
        else
            hipLaunchKernelGGL(sample_max_kernel_f32<2>, sgrid, dim3(256), 0, st, U, V, ld, n_items, user_ids, n_b, g.b_pad, n_s_tiles, tpg, n_groups, gmax, garg);
        QREC_LAUNCH_CHECK();
        const unsigned lane_blocks = (unsigned)((n_b + 255) / 256);
        hipLaunchKernelGGL(threshold_var_kernel, dim3(lane_blocks), dim3(256), 0, st, gmax, garg, user_ids, rated_indptr, rated_sorted, g.b_pad, n_b,
                           n_groups, M, tau);
        QREC_LAUNCH_CHECK();
        if (use_bf16) {
            hipLaunchKernelGGL(lowered_tau2_kernel, dim3(lane_blocks), dim3(256), 0, st, tau, u_norm, v_max, n_b, tau_low);
            QREC_LAUNCH_CHECK();
        }
    }
    // (B) score + filter, (C) select
    QREC_HIP_CHECK(hipMemsetAsync(n_flagged, 0, sizeof(int32_t), st));
    // measured at the Yelp2018 shape (kernel time of this pass; 0.98 ms of pure MFMA time): one user tile per wavefront at 4
    // wavefronts per SIMD 2.49 ms (twice the operand loads per MFMA); two tiles at one wavefront per SIMD 1.72, at two
    // wavefronts per SIMD 1.67 (launched here); the item tile shared by a block's wavefronts through LDS (coalesced fetch,
    // one barrier per tile) 1.89; the epilogue software-pipelined under the next tile's MFMAs over two accumulator pairs
    // 1.92 (448 registers, accumulators shuttling between AGPRs and VGPRs) although the same loop shape gains 22 % in
    // tools/ubench/mfma_tile.hip; thresholds at +inf (no element ever appended) 1.66: the pass is bound by the per-lane-row
    // operand fetch interleaved with the accumulator read-out, not by the compare / append work
    const dim3 grid((unsigned)g.grid_x, (unsigned)g.grid_y);
    if (use_bf16) {
        // (B') the filter on v_mfma_f32_32x32x16_bf16 against the lowered thresholds
#define QREC_BF(NM, NU)                                                                                                               \
        hipLaunchKernelGGL((score_filter_bf16_kernel<NM, NU>), grid, dim3(256), 0, st, Ub, (int)g.b_pad, Vb, n_items, n_b, tau_low, g.n_lists,     \
                           g.list_cap, cand_s, cand_i, cand_n)
        if (ld == 32) { QREC_BF(2, 4); }
        else if (ld == 64) { if (g.nu == 2) { QREC_BF(4, 2); } else { QREC_BF(4, 4); } }
        else { QREC_BF(8, 2); }
#undef QREC_BF
    } else if (ld <= 64)
        hipLaunchKernelGGL((score_filter2_kernel_f32<1, 2>), grid, dim3(256), 0, st, U, V, ld, n_items, user_ids, n_b, 0, tau, g.n_lists,
                           cand_s, cand_i, cand_n);
    else
        hipLaunchKernelGGL((score_filter2_kernel_f32<2, 1>), grid, dim3(256), 0, st, U, V, ld, n_items, user_ids, n_b, 0, tau, g.n_lists,
                           cand_s, cand_i, cand_n);
    QREC_LAUNCH_CHECK();
    // candidates per user: ~ (N + 1) * kSampleStride * 1.1, times ~1.4 behind the bf16 filter; the pool holds 4x that
    int pool_cap = g.n_lists * g.list_cap;
    const int want = 4 * (int)((K + 1) * (use_bf16 ? bf16_stride : kSampleStride) * 1.6);
    if (pool_cap > want) pool_cap = want < 512 ? 512 : want;      // trained tables (uneven item norms) loosen the bound: keep room
    const size_t lds = (size_t)kSelectWaves * pool_cap * sizeof(unsigned long long);
    const dim3 sgrid((unsigned)((n_b + kSelectWaves - 1) / kSelectWaves));
    const float *Ur = use_bf16 ? U : (const float *)nullptr;
    if (ld <= 64) {
        QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&select_topk_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(select_topk_kernel<1>, sgrid, dim3(64 * kSelectWaves), lds, st, cand_s, cand_i, cand_n, g.n_lists, tau, user_ids, rated_indptr,
                           rated_sorted, n_b, K, ids_out, scores_out, flags, n_flagged, flagged_list, Ur, V, ld, pool_cap,
                           reinterpret_cast<const float *>(base + g.off_taul), g.list_cap);
    } else {
        QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&select_topk_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(select_topk_kernel<2>, sgrid, dim3(64 * kSelectWaves), lds, st, cand_s, cand_i, cand_n, g.n_lists, tau, user_ids, rated_indptr,
                           rated_sorted, n_b, K, ids_out, scores_out, flags, n_flagged, flagged_list, Ur, V, ld, pool_cap,
                           reinterpret_cast<const float *>(base + g.off_taul), g.list_cap);
    }
    QREC_LAUNCH_CHECK();
    // (D) users whose heap history matters: the block path, fb_users at a time
    int32_t h_nf = 0;
    QREC_HIP_CHECK(hipMemcpyAsync(&h_nf, n_flagged, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    QREC_HIP_CHECK(hipStreamSynchronize(st));
    int32_t *fb_users = reinterpret_cast<int32_t *>(base + g.off_fbu), *fb_ids = reinterpret_cast<int32_t *>(base + g.off_fbi);
    float *fb_sc = reinterpret_cast<float *>(base + g.off_fbs);
    // Their scores as USER-MAJOR rows ([user][item]: a step of the walk -- 64 consecutive items -- is one 256-B access), masked,
    // then the exact sequential emulation, one wavefront per user: three launches.  (History: through the whole block route --
    // sliced top-N, flags, exact_wave_kernel on an [item][b_pad] block, every load of a step its own cache line -- the fallback
    // was 0.8 ms of the evaluation's 3.05; as [panel][item][64] panels, a column striding 256 B, 0.18 ms for a handful of users.)
    float *fb_block = reinterpret_cast<float *>(base + g.off_fb);
    const int64_t row_len = (int64_t)g.n_item_tiles * 32, panel = row_len * 64;
    for (int off = 0; off < h_nf; off += g.fb_users) {
        const int n = h_nf - off < g.fb_users ? h_nf - off : g.fb_users;
        hipLaunchKernelGGL(gather_user_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, user_ids, flagged_list, off, n, fb_users);
        QREC_LAUNCH_CHECK();
        const int n_utiles = (n + 63) / 64;
        int splits = (4096 + n_utiles - 1) / n_utiles;
        int per_wave = (g.n_item_tiles + splits - 1) / splits;
        if (per_wave < 2) per_wave = g.n_item_tiles < 2 ? g.n_item_tiles : 2;      // a handful of users: many short wavefronts
        const int waves_per_utile = (g.n_item_tiles + per_wave - 1) / per_wave;
        // user-major rows (see score_kernel_f32): item stride 1, user stride row_len
        hipLaunchKernelGGL(score_kernel_f32, dim3((unsigned)n_utiles, (unsigned)((waves_per_utile + 3) / 4)), dim3(256), 0, st, U, V, ld,
                           n_items, fb_users, n, 1, per_wave, fb_block, 1, panel, row_len);
        QREC_LAUNCH_CHECK();
        if (rated_indptr) {
            hipLaunchKernelGGL(mask_kernel<float>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, fb_users, n, rated_indptr, rated_sorted,
                               1, fb_block, 32, 1, panel, row_len);
            QREC_LAUNCH_CHECK();
        }
        {                                                         // fused route: K + 1 <= 64
            const int stage_items = n_items < kWalkChunk ? (n_items + 3) / 4 * 4 : kWalkChunk;
            const size_t walk_lds = (size_t)stage_items * sizeof(float);
            QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&exact_walk_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds));
            hipLaunchKernelGGL(exact_walk_lds_kernel, dim3((unsigned)n), dim3(256), walk_lds, st, fb_block, row_len, n_items, K, fb_ids, fb_sc);
        }
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n * K + 255) / 256)), dim3(256), 0, st, flagged_list, off, n, K, fb_ids, fb_sc,
                           ids_out, scores_out);
        QREC_LAUNCH_CHECK();
    }
    return QREC_OK;
}

// Measure.hits (util/measure.py:15-21) and the DCG sum of Measure.NDCG (util/measure.py:70-82) for the lists a
// previous qrec_score_topk left on the device: one lane per user walks its first n_cut recommendations in rank
// order, looks each id up in the user's sorted test items and adds the caller's discount[pos] (the host passes
// Python's own 1/math.log(pos+2) doubles) in that order -- the same sequential fp64 sum the reference runs.
__global__ __launch_bounds__(256) void rank_hits_kernel(const int32_t *__restrict__ ids, int n_users, int row_stride,
                                                        int n_cut, const int32_t *__restrict__ user_ids,
                                                        const int64_t *__restrict__ test_indptr,
                                                        const int32_t *__restrict__ test_items,
                                                        const double *__restrict__ discount,
                                                        int32_t *__restrict__ hits_out, double *__restrict__ dcg_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_users) return;
    const int u = user_ids[b];
    const int64_t lo0 = test_indptr[u], hi0 = test_indptr[u + 1];
    int hits = 0;
    double dcg = 0.0;
    for (int pos = 0; pos < n_cut; pos++) {
        const int item = ids[(int64_t)b * row_stride + pos];
        if (item < 0) break;                       // -1 padding: fewer than N items exist
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (test_items[mid] < item) lo = mid + 1; else hi = mid;
        }
        if (lo < hi0 && test_items[lo] == item) { hits++; dcg += discount[pos]; }
    }
    hits_out[b] = hits;
    dcg_out[b] = dcg;
}

}  // namespace

extern "C" {

int qrec_score_topk_scratch_bytes(int dtype, int32_t n_items, int32_t n_batch_users, int32_t ld, int32_t K, int64_t *bytes) {
    QREC_REQUIRE(bytes && n_items >= 0 && n_batch_users >= 0, "qrec_score_topk_scratch_bytes: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_score_topk_scratch_bytes: bad dtype %d", dtype);
    if (fused_ok(dtype, ld, n_items, K) && n_batch_users > 0) {
        // fused path: the sampled block (1/8), candidate lists, and a block-path scratch for the flagged users' rounds
        *bytes = (int64_t)fused_geometry(n_items, n_batch_users, ld).total;
        return QREC_OK;
    }
    // the transposed score block, then the sliced top-N's candidates (scores + ids), flags, flagged list, counter
    *bytes = (int64_t)block_path_bytes(dtype == QREC_F64 ? 8 : 4, n_items, n_batch_users);
    return QREC_OK;
}

int qrec_score_topk(const void *d_U, const void *d_V, int dtype, int32_t d, int32_t ld, int32_t n_items,
                    const int32_t *d_user_ids, int32_t n_batch_users, const int64_t *d_rated_indptr,
                    const int32_t *d_rated_items, int32_t K, void *d_scratch, int32_t *d_ids_out,
                    void *d_scores_out, void *stream) {
    QREC_REQUIRE(d_U && d_V && d_user_ids && d_scratch && d_ids_out && d_scores_out, "qrec_score_topk: null argument");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_score_topk: bad dtype %d", dtype);
    QREC_REQUIRE(d >= 1 && ld >= d && n_items >= 1 && n_batch_users >= 0, "qrec_score_topk: bad sizes");
    QREC_REQUIRE(ld % (dtype == QREC_F64 ? 16 : 32) == 0,
                 "qrec_score_topk: the row stride must be a multiple of 32 floats / 16 doubles, pad columns zero (got ld=%d)", ld);
    QREC_REQUIRE(K >= 1 && K <= 100, "qrec_score_topk: N must be in 1..100 (base/recommender.py:132-134)");
    QREC_REQUIRE((d_rated_indptr == nullptr) == (d_rated_items == nullptr), "qrec_score_topk: rated CSR incomplete");
    if (n_batch_users == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    if (fused_ok(dtype, ld, n_items, K))
        return run_fused_topk_f32((const float *)d_U, (const float *)d_V, d, ld, n_items, d_user_ids, n_batch_users, d_rated_indptr,
                                  d_rated_items, K, d_scratch, d_ids_out, (float *)d_scores_out, st);
    return dtype == QREC_F64
               ? run_score_topk<double>(d_U, d_V, d, ld, n_items, d_user_ids, n_batch_users, d_rated_indptr,
                                        d_rated_items, K, d_scratch, d_ids_out, d_scores_out, st)
               : run_score_topk<float>(d_U, d_V, d, ld, n_items, d_user_ids, n_batch_users, d_rated_indptr,
                                       d_rated_items, K, d_scratch, d_ids_out, d_scores_out, st);
}

int qrec_rank_hits(const int32_t *d_ids, int32_t n_batch_users, int32_t row_stride, int32_t n_cut,
                   const int32_t *d_user_ids, const int64_t *d_test_indptr, const int32_t *d_test_items,
                   const double *d_discount, int32_t *d_hits_out, double *d_dcg_out, void *stream) {
    QREC_REQUIRE(d_ids && d_user_ids && d_test_indptr && d_discount && d_hits_out && d_dcg_out, "qrec_rank_hits: null argument");
    QREC_REQUIRE(n_batch_users >= 0 && n_cut >= 1 && row_stride >= n_cut, "qrec_rank_hits: bad sizes");
    if (n_batch_users == 0) return QREC_OK;
    hipLaunchKernelGGL(rank_hits_kernel, dim3((unsigned)((n_batch_users + 255) / 256)), dim3(256), 0, as_stream(stream), d_ids,
                       n_batch_users, row_stride, n_cut, d_user_ids, d_test_indptr, d_test_items, d_discount, d_hits_out,
                       d_dcg_out);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
