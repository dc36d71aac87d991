// Device side of the multi-GPU BPR path: what runs between the SGD kernel and the collectives.
//
// Replicated item table:  delta = Q - Q_start  ->  all-reduce(delta)  ->  Q_start += delta; Q = Q_start.
// Row-sharded item table (item r*G+o lives on rank o as local row r): per batch the requester
//   (1) builds the list of DISTINCT item rows its triplets touch, grouped by owner, and rewrites the triplets' item
//       ids into slots of a local row cache (plan kernels below);
//   (2) owners gather the requested rows (gather_rows_kernel) and ship them (qrec_alltoall_rows);
//   (3) the unchanged Hogwild SGD kernel runs on (P_local, cache);
//   (4) the cache travels back and owners add  returned - sent  into their rows (row_delta_kernel, f32 atomics: several
//       ranks may return the same row).
// All of it is HBM streaming / gather work: float4 per lane, one contiguous 64-B+ segment per group of lanes.
#include "common.h"

using namespace qrec;

namespace {

constexpr int kTile = 1024;      // positions per plan tile (256 threads x 4)
constexpr int kPlanBlock = 256;

__global__ void table_delta_kernel(const float4 *__restrict__ table, const float4 *__restrict__ start,
                                   float4 *__restrict__ delta, int64_t n4) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = table[k], b = start[k];
        delta[k] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
}

__global__ void table_apply_kernel(float4 *__restrict__ table, float4 *__restrict__ start,
                                   const float4 *__restrict__ delta, int64_t n4) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        const float4 s = start[k], d = delta[k];
        const float4 r = make_float4(s.x + d.x, s.y + d.y, s.z + d.z, s.w + d.w);
        start[k] = r;
        table[k] = r;
    }
}

// ---- plan of one batch -------------------------------------------------------------------------------------------
// position of an item in owner-major order; rows_pad = rows per owner rounded up to a whole number of tiles, so a
// tile never straddles two owners
__device__ inline int64_t item_pos(int32_t item, int32_t world, int64_t rows_pad) {
    return (int64_t)(item % world) * rows_pad + item / world;
}

__global__ void plan_mark_kernel(const int32_t *__restrict__ i, const int32_t *__restrict__ j, int64_t n, int32_t world,
                                 int64_t rows_pad, int32_t *__restrict__ flags) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        flags[item_pos(i[t], world, rows_pad)] = 1;
        flags[item_pos(j[t], world, rows_pad)] = 1;
    }
}

__device__ inline int block_exclusive_scan(int v, int *lds, int *total) {
    // 256 threads = 4 wavefronts: inclusive scan inside the wavefront, wavefront totals through LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(inc, s, 64);
        if (lane >= s) inc += o;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += lds[k];
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}

__global__ void plan_count_kernel(const int32_t *__restrict__ flags, int32_t *__restrict__ tile_cnt) {
    __shared__ int lds[4];
    const int4 f = reinterpret_cast<const int4 *>(flags)[blockIdx.x * (int64_t)kPlanBlock + threadIdx.x];
    int total;
    block_exclusive_scan((f.x != 0) + (f.y != 0) + (f.z != 0) + (f.w != 0), lds, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// one block: tile_cnt[0..ntiles) -> exclusive offsets in place, tile_cnt[ntiles] = total, counts[o] per owner
__global__ void plan_scan_kernel(int32_t *__restrict__ tile_cnt, int32_t ntiles, int32_t tiles_per_owner, int32_t world,
                                 int32_t *__restrict__ counts) {
    __shared__ int lds[4];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += kPlanBlock) {
        const int k = base + threadIdx.x;
        const int v = k < ntiles ? tile_cnt[k] : 0;
        int total;
        const int ex = block_exclusive_scan(v, lds, &total);
        const int c = carry;
        if (k < ntiles) tile_cnt[k] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_cnt[ntiles] = carry;
    __syncthreads();
    for (int o = threadIdx.x; o < world; o += blockDim.x)
        counts[o] = tile_cnt[(o + 1) * tiles_per_owner] - tile_cnt[o * tiles_per_owner];
}

__global__ void plan_fill_kernel(int32_t *__restrict__ flags, const int32_t *__restrict__ tile_off, int64_t rows_pad,
                                 int32_t *__restrict__ req_rows) {
    __shared__ int lds[4];
    const int64_t p4 = blockIdx.x * (int64_t)kPlanBlock + threadIdx.x;
    int4 f = reinterpret_cast<int4 *>(flags)[p4];
    int total;
    int slot = tile_off[blockIdx.x] + block_exclusive_scan((f.x != 0) + (f.y != 0) + (f.z != 0) + (f.w != 0), lds, &total);
    if (f.x | f.y | f.z | f.w) {
        const int64_t pos = p4 * 4;
        const int32_t row0 = (int32_t)(pos % rows_pad);     // local row at the owner (a tile lies inside one owner)
        if (f.x) { req_rows[slot] = row0; f.x = ++slot; }
        if (f.y) { req_rows[slot] = row0 + 1; f.y = ++slot; }
        if (f.z) { req_rows[slot] = row0 + 2; f.z = ++slot; }
        if (f.w) { req_rows[slot] = row0 + 3; f.w = ++slot; }
        reinterpret_cast<int4 *>(flags)[p4] = f;            // flag -> slot + 1
    }
}

__global__ void plan_remap_kernel(const int32_t *__restrict__ i, const int32_t *__restrict__ j, int64_t n, int32_t world,
                                  int64_t rows_pad, const int32_t *__restrict__ flags, int32_t *__restrict__ ci,
                                  int32_t *__restrict__ cj) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        ci[t] = flags[item_pos(i[t], world, rows_pad)] - 1;
        cj[t] = flags[item_pos(j[t], world, rows_pad)] - 1;
    }
}

// ---- the same plan for ALL batches of an epoch in one set of launches (blockIdx.y / blockIdx.x = batch) ---------------------
// batch b covers the triplets [bounds[b], bounds[b + 1]); its flags / tile counters live at b * stride in the scratch, its request
// list starts at req_off[b], its counts at counts[b * world].  Five launches per epoch instead of five per batch.
__global__ void plan_mark_epoch_kernel(const int32_t *__restrict__ i, const int32_t *__restrict__ j, const int64_t *__restrict__ bounds,
                                       int32_t world, int64_t rows_pad, int32_t *__restrict__ flags, int64_t flag_stride) {
    const int b = blockIdx.y;
    const int64_t t0 = bounds[b], t1 = bounds[b + 1];
    int32_t *f = flags + b * flag_stride;
    for (int64_t t = t0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < t1; t += (int64_t)gridDim.x * blockDim.x) {
        f[item_pos(i[t], world, rows_pad)] = 1;
        f[item_pos(j[t], world, rows_pad)] = 1;
    }
}

__global__ void plan_count_epoch_kernel(const int32_t *__restrict__ flags, int64_t flag_stride, int32_t *__restrict__ tile_cnt,
                                        int64_t tile_stride) {
    __shared__ int lds[4];
    const int b = blockIdx.y;
    const int4 f = reinterpret_cast<const int4 *>(flags + b * flag_stride)[blockIdx.x * (int64_t)kPlanBlock + threadIdx.x];
    int total;
    block_exclusive_scan((f.x != 0) + (f.y != 0) + (f.z != 0) + (f.w != 0), lds, &total);
    if (threadIdx.x == 0) tile_cnt[b * tile_stride + blockIdx.x] = total;
}

__global__ void plan_scan_epoch_kernel(int32_t *__restrict__ tile_cnt_all, int64_t tile_stride, int32_t ntiles, int32_t tiles_per_owner,
                                       int32_t world, int32_t *__restrict__ counts_all) {
    __shared__ int lds[4];
    __shared__ int carry;
    int32_t *tile_cnt = tile_cnt_all + blockIdx.x * tile_stride;
    int32_t *counts = counts_all + blockIdx.x * world;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += kPlanBlock) {
        const int k = base + threadIdx.x;
        const int v = k < ntiles ? tile_cnt[k] : 0;
        int total;
        const int ex = block_exclusive_scan(v, lds, &total);
        const int c = carry;
        if (k < ntiles) tile_cnt[k] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_cnt[ntiles] = carry;
    __syncthreads();
    for (int o = threadIdx.x; o < world; o += blockDim.x)
        counts[o] = tile_cnt[(o + 1) * tiles_per_owner] - tile_cnt[o * tiles_per_owner];
}

__global__ void plan_fill_epoch_kernel(int32_t *__restrict__ flags_all, int64_t flag_stride, const int32_t *__restrict__ tile_all,
                                       int64_t tile_stride, int64_t rows_pad, int32_t *__restrict__ req_rows,
                                       const int64_t *__restrict__ req_off) {
    __shared__ int lds[4];
    const int b = blockIdx.y;
    int32_t *flags = flags_all + b * flag_stride;
    int32_t *req = req_rows + req_off[b];
    const int64_t p4 = blockIdx.x * (int64_t)kPlanBlock + threadIdx.x;
    int4 f = reinterpret_cast<int4 *>(flags)[p4];
    int total;
    int slot = tile_all[b * tile_stride + blockIdx.x] + block_exclusive_scan((f.x != 0) + (f.y != 0) + (f.z != 0) + (f.w != 0), lds, &total);
    if (f.x | f.y | f.z | f.w) {
        const int64_t pos = p4 * 4;
        const int32_t row0 = (int32_t)(pos % rows_pad);
        if (f.x) { req[slot] = row0; f.x = ++slot; }
        if (f.y) { req[slot] = row0 + 1; f.y = ++slot; }
        if (f.z) { req[slot] = row0 + 2; f.z = ++slot; }
        if (f.w) { req[slot] = row0 + 3; f.w = ++slot; }
        reinterpret_cast<int4 *>(flags)[p4] = f;
    }
}

__global__ void plan_remap_epoch_kernel(const int32_t *__restrict__ i, const int32_t *__restrict__ j, const int64_t *__restrict__ bounds,
                                        int32_t world, int64_t rows_pad, const int32_t *__restrict__ flags_all, int64_t flag_stride,
                                        int32_t *__restrict__ ci, int32_t *__restrict__ cj) {
    const int b = blockIdx.y;
    const int64_t t0 = bounds[b], t1 = bounds[b + 1];
    const int32_t *flags = flags_all + b * flag_stride;
    for (int64_t t = t0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < t1; t += (int64_t)gridDim.x * blockDim.x) {
        ci[t] = flags[item_pos(i[t], world, rows_pad)] - 1;
        cj[t] = flags[item_pos(j[t], world, rows_pad)] - 1;
    }
}

// ---- rows in and out of a shard ----------------------------------------------------------------------------------
template <int LD4>   // float4s per row
__global__ void gather_rows_kernel(const float4 *__restrict__ table, const int32_t *__restrict__ rows, int64_t n,
                                   float4 *__restrict__ out) {
    const int64_t total = n * LD4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t / LD4;
        const int c = (int)(t % LD4);
        out[t] = table[(int64_t)rows[k] * LD4 + c];
    }
}

template <int LD>    // floats per row; one thread per float so that an atomic instruction covers contiguous 256 B
__global__ void row_delta_kernel(float *__restrict__ table, const int32_t *__restrict__ rows, int64_t n,
                                 const float *__restrict__ fresh, const float *__restrict__ sent) {
    const int64_t total = n * LD;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const float dlt = fresh[t] - sent[t];
        if (dlt != 0.0f) atomicAdd(table + (int64_t)rows[t / LD] * LD + (t % LD), dlt);
    }
}

// ---- the rows of a training batch out of / into row-partitioned tables (graph models, round 4) -------------------------------
// Batch row k of 3B: k < B -> table row u[k];  k < 2B -> n_users + i[k - B];  else n_users + j[k - 2B]  (the rows embedding_lookup
// reads, LightGCN.py:22-24).  Gather: out[k] = the row if this rank's block [lo, hi) holds it, else zeros -- the sum over the ranks of
// these 3B-row tables IS the batch's rows of the whole table (one small all-reduce instead of an all-gather of the table).
// Scatter: table[row(k)] += src[k] for every k whose row lies in [lo, hi) (f32 atomics: a batch repeats rows).
__device__ __forceinline__ int64_t batch_row_id(const int32_t *u, const int32_t *i, const int32_t *j, int B, int64_t n_users, int64_t k) {
    return k < B ? (int64_t)u[k] : (k < 2 * (int64_t)B ? n_users + i[k - B] : n_users + j[k - 2 * (int64_t)B]);
}
template <int LD4>
__global__ void batch_rows_gather_kernel(const float4 *__restrict__ block, int64_t lo, int64_t hi, const int32_t *__restrict__ u,
                                         const int32_t *__restrict__ i, const int32_t *__restrict__ j, int B, int64_t n_users,
                                         float4 *__restrict__ out) {
    const int64_t total = 3 * (int64_t)B * LD4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t / LD4, row = batch_row_id(u, i, j, B, n_users, k);
        out[t] = (row >= lo && row < hi) ? block[(row - lo) * LD4 + (t % LD4)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int LD>
__global__ void batch_rows_scatter_kernel(float *__restrict__ block, int64_t lo, int64_t hi, const int32_t *__restrict__ u,
                                          const int32_t *__restrict__ i, const int32_t *__restrict__ j, int B, int64_t n_users,
                                          const float *__restrict__ src) {
    const int64_t total = 3 * (int64_t)B * LD;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = batch_row_id(u, i, j, B, n_users, t / LD);
        const float v = src[t];
        if (row >= lo && row < hi && v != 0.0f) atomicAdd(block + (row - lo) * LD + (t % LD), v);
    }
}

// the same two operations for an arbitrary list of table rows (SimGCL's InfoNCE reads the batch's UNIQUE users / items, SimGCL.py:61-64)
template <int LD4>
__global__ void rows_gather_owned_kernel(const float4 *__restrict__ block, int64_t lo, int64_t hi, const int32_t *__restrict__ ids, int64_t n,
                                         float4 *__restrict__ out) {
    const int64_t total = n * LD4;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = ids[t / LD4];
        out[t] = (row >= lo && row < hi) ? block[(row - lo) * LD4 + (t % LD4)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int LD>
__global__ void rows_scatter_owned_kernel(float *__restrict__ block, int64_t lo, int64_t hi, const int32_t *__restrict__ ids, int64_t n,
                                          const float *__restrict__ src) {
    const int64_t total = n * LD;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = ids[t / LD];
        const float v = src[t];
        if (row >= lo && row < hi && v != 0.0f) atomicAdd(block + (row - lo) * LD + (t % LD), v);
    }
}

inline int grid_for(int64_t work, int block) {
    int64_t g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

inline int64_t rows_pad_of(int64_t n_items, int32_t world) {
    const int64_t rows = (n_items + world - 1) / world;
    return (rows + kTile - 1) / kTile * kTile;
}

}  // namespace

extern "C" {

int qrec_table_delta(const float *d_table, const float *d_start, float *d_delta, int64_t n, void *stream) {
    QREC_REQUIRE(n >= 0 && n % 4 == 0, "qrec_table_delta: element count must be a multiple of 4 (got %lld)", (long long)n);
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_table && d_start && d_delta, "qrec_table_delta: null argument");
    hipLaunchKernelGGL(table_delta_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, as_stream(stream),
                       (const float4 *)d_table, (const float4 *)d_start, (float4 *)d_delta, n / 4);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_table_apply(float *d_table, float *d_start, const float *d_delta, int64_t n, void *stream) {
    QREC_REQUIRE(n >= 0 && n % 4 == 0, "qrec_table_apply: element count must be a multiple of 4 (got %lld)", (long long)n);
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_table && d_start && d_delta, "qrec_table_apply: null argument");
    hipLaunchKernelGGL(table_apply_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, as_stream(stream), (float4 *)d_table,
                       (float4 *)d_start, (const float4 *)d_delta, n / 4);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_shard_rows(int64_t n_items, int32_t world, int32_t rank, int64_t *rows) {
    QREC_REQUIRE(rows && world >= 1 && rank >= 0 && rank < world && n_items >= 0, "qrec_shard_rows: bad arguments");
    *rows = n_items / world + (rank < n_items % world ? 1 : 0);
    return QREC_OK;
}

int qrec_shard_plan_scratch_bytes(int64_t n_items, int32_t world, int64_t *bytes) {
    QREC_REQUIRE(bytes && world >= 1 && n_items >= 1, "qrec_shard_plan_scratch_bytes: bad arguments");
    const int64_t rows_pad = rows_pad_of(n_items, world), npos = rows_pad * world;
    QREC_REQUIRE(npos / kTile < (1 << 30), "qrec_shard_plan_scratch_bytes: catalogue too large");
    *bytes = npos * 4 + (npos / kTile + 1 + 3) / 4 * 16;
    return QREC_OK;
}

int qrec_shard_plan_batch(const int32_t *d_i, const int32_t *d_j, int64_t n, int64_t n_items, int32_t world,
                          void *d_scratch, int32_t *d_req_rows, int32_t *d_counts, int32_t *d_ci, int32_t *d_cj,
                          void *stream) {
    QREC_REQUIRE(world >= 1 && n_items >= 1 && n >= 0, "qrec_shard_plan_batch: bad sizes");
    QREC_REQUIRE(d_scratch && d_counts, "qrec_shard_plan_batch: null argument");
    QREC_REQUIRE(n == 0 || (d_i && d_j && d_req_rows && d_ci && d_cj), "qrec_shard_plan_batch: null array");
    hipStream_t st = as_stream(stream);
    const int64_t rows_pad = rows_pad_of(n_items, world), npos = rows_pad * world;
    const int32_t ntiles = (int32_t)(npos / kTile), tpo = (int32_t)(rows_pad / kTile);
    int32_t *flags = static_cast<int32_t *>(d_scratch), *tile = flags + npos;
    QREC_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)npos * 4, st));
    if (n) hipLaunchKernelGGL(plan_mark_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, d_i, d_j, n, world, rows_pad, flags);
    hipLaunchKernelGGL(plan_count_kernel, dim3(ntiles), dim3(kPlanBlock), 0, st, flags, tile);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(kPlanBlock), 0, st, tile, ntiles, tpo, world, d_counts);
    if (n) {
        hipLaunchKernelGGL(plan_fill_kernel, dim3(ntiles), dim3(kPlanBlock), 0, st, flags, tile, rows_pad, d_req_rows);
        hipLaunchKernelGGL(plan_remap_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, d_i, d_j, n, world, rows_pad, flags,
                           d_ci, d_cj);
    }
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_shard_plan_epoch_scratch_bytes(int64_t n_items, int32_t world, int32_t n_batches, int64_t *bytes) {
    QREC_REQUIRE(bytes && world >= 1 && n_items >= 1 && n_batches >= 1 && n_batches <= 65535, "qrec_shard_plan_epoch_scratch_bytes: bad arguments");
    int64_t one = 0;
    int rc = qrec_shard_plan_scratch_bytes(n_items, world, &one);
    if (rc != QREC_OK) return rc;
    *bytes = one * n_batches;
    return QREC_OK;
}

int qrec_shard_plan_epoch(const int32_t *d_i, const int32_t *d_j, const int64_t *d_bounds, int32_t n_batches, int64_t n,
                          int64_t n_items, int32_t world, void *d_scratch, int32_t *d_req_rows, const int64_t *d_req_off,
                          int32_t *d_counts, int32_t *d_ci, int32_t *d_cj, void *stream) {
    QREC_REQUIRE(world >= 1 && n_items >= 1 && n >= 0 && n_batches >= 1 && n_batches <= 65535, "qrec_shard_plan_epoch: bad sizes");
    QREC_REQUIRE(d_scratch && d_counts && d_bounds && d_req_off, "qrec_shard_plan_epoch: null argument");
    QREC_REQUIRE(n == 0 || (d_i && d_j && d_req_rows && d_ci && d_cj), "qrec_shard_plan_epoch: null array");
    hipStream_t st = as_stream(stream);
    const int64_t rows_pad = rows_pad_of(n_items, world), npos = rows_pad * world;
    const int32_t ntiles = (int32_t)(npos / kTile), tpo = (int32_t)(rows_pad / kTile);
    int64_t one = 0;
    qrec_shard_plan_scratch_bytes(n_items, world, &one);
    // scratch of batch b: [npos flags][ntiles + 1 tile counters, padded], `one` bytes each, back to back
    const int64_t stride_words = one / 4;
    int32_t *flags = static_cast<int32_t *>(d_scratch), *tile = flags + npos;
    QREC_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)one * n_batches, st));
    const int gx = grid_for((n + n_batches - 1) / n_batches, 256);
    if (n) hipLaunchKernelGGL(plan_mark_epoch_kernel, dim3(gx, n_batches), dim3(256), 0, st, d_i, d_j, d_bounds, world, rows_pad, flags, stride_words);
    hipLaunchKernelGGL(plan_count_epoch_kernel, dim3(ntiles, n_batches), dim3(kPlanBlock), 0, st, flags, stride_words, tile, stride_words);
    hipLaunchKernelGGL(plan_scan_epoch_kernel, dim3(n_batches), dim3(kPlanBlock), 0, st, tile, stride_words, ntiles, tpo, world, d_counts);
    if (n) {
        hipLaunchKernelGGL(plan_fill_epoch_kernel, dim3(ntiles, n_batches), dim3(kPlanBlock), 0, st, flags, stride_words, tile, stride_words,
                           rows_pad, d_req_rows, d_req_off);
        hipLaunchKernelGGL(plan_remap_epoch_kernel, dim3(gx, n_batches), dim3(256), 0, st, d_i, d_j, d_bounds, world, rows_pad, flags,
                           stride_words, d_ci, d_cj);
    }
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_gather_rows(const float *d_table, int32_t ld, const int32_t *d_rows, int64_t n, float *d_out, void *stream) {
    QREC_REQUIRE(n >= 0, "qrec_gather_rows: negative count");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_gather_rows: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_table && d_rows && d_out, "qrec_gather_rows: null argument");
    hipStream_t st = as_stream(stream);
    const dim3 g(grid_for(n * (ld / 4), 256)), b(256);
#define QREC_GATHER(L4) hipLaunchKernelGGL((gather_rows_kernel<L4>), g, b, 0, st, (const float4 *)d_table, d_rows, n, (float4 *)d_out)
    switch (ld) {
        case 32: QREC_GATHER(8); break;
        case 64: QREC_GATHER(16); break;
        case 128: QREC_GATHER(32); break;
        default: QREC_GATHER(64); break;
    }
#undef QREC_GATHER
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_scatter_add_row_deltas(float *d_table, int32_t ld, const int32_t *d_rows, int64_t n, const float *d_fresh,
                                const float *d_sent, void *stream) {
    QREC_REQUIRE(n >= 0, "qrec_scatter_add_row_deltas: negative count");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_scatter_add_row_deltas: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_table && d_rows && d_fresh && d_sent, "qrec_scatter_add_row_deltas: null argument");
    hipStream_t st = as_stream(stream);
    const dim3 g(grid_for(n * ld, 256)), b(256);
#define QREC_DELTA(L) hipLaunchKernelGGL((row_delta_kernel<L>), g, b, 0, st, d_table, d_rows, n, d_fresh, d_sent)
    switch (ld) {
        case 32: QREC_DELTA(32); break;
        case 64: QREC_DELTA(64); break;
        case 128: QREC_DELTA(128); break;
        default: QREC_DELTA(256); break;
    }
#undef QREC_DELTA
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_rows_gather_owned(const float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_ids, int64_t n, float *d_out, void *stream) {
    QREC_REQUIRE(n >= 0 && lo >= 0 && hi >= lo, "qrec_rows_gather_owned: bad arguments");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_rows_gather_owned: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_block && d_ids && d_out, "qrec_rows_gather_owned: null argument");
    const dim3 g(grid_for(n * (ld / 4), 256)), b(256);
#define QREC_RG(L4) hipLaunchKernelGGL((rows_gather_owned_kernel<L4>), g, b, 0, as_stream(stream), (const float4 *)d_block, lo, hi, d_ids, n, (float4 *)d_out)
    switch (ld) {
        case 32: QREC_RG(8); break;
        case 64: QREC_RG(16); break;
        case 128: QREC_RG(32); break;
        default: QREC_RG(64); break;
    }
#undef QREC_RG
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_rows_scatter_add_owned(float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_ids, int64_t n, const float *d_src, void *stream) {
    QREC_REQUIRE(n >= 0 && lo >= 0 && hi >= lo, "qrec_rows_scatter_add_owned: bad arguments");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_rows_scatter_add_owned: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_block && d_ids && d_src, "qrec_rows_scatter_add_owned: null argument");
    const dim3 g(grid_for(n * ld, 256)), b(256);
#define QREC_RS(L) hipLaunchKernelGGL((rows_scatter_owned_kernel<L>), g, b, 0, as_stream(stream), d_block, lo, hi, d_ids, n, d_src)
    switch (ld) {
        case 32: QREC_RS(32); break;
        case 64: QREC_RS(64); break;
        case 128: QREC_RS(128); break;
        default: QREC_RS(256); break;
    }
#undef QREC_RS
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_batch_rows_gather(const float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_u, const int32_t *d_i,
                           const int32_t *d_j, int32_t B, int64_t n_users, float *d_out, void *stream) {
    QREC_REQUIRE(B >= 0 && lo >= 0 && hi >= lo && n_users >= 0, "qrec_batch_rows_gather: bad arguments");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_batch_rows_gather: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (B == 0) return QREC_OK;
    QREC_REQUIRE(d_block && d_u && d_i && d_j && d_out, "qrec_batch_rows_gather: null argument");
    const dim3 g(grid_for(3 * (int64_t)B * (ld / 4), 256)), b(256);
#define QREC_BG(L4) hipLaunchKernelGGL((batch_rows_gather_kernel<L4>), g, b, 0, as_stream(stream), (const float4 *)d_block, lo, hi, d_u, d_i, d_j, B, n_users, (float4 *)d_out)
    switch (ld) {
        case 32: QREC_BG(8); break;
        case 64: QREC_BG(16); break;
        case 128: QREC_BG(32); break;
        default: QREC_BG(64); break;
    }
#undef QREC_BG
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_batch_rows_scatter_add(float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_u, const int32_t *d_i,
                                const int32_t *d_j, int32_t B, int64_t n_users, const float *d_src, void *stream) {
    QREC_REQUIRE(B >= 0 && lo >= 0 && hi >= lo && n_users >= 0, "qrec_batch_rows_scatter_add: bad arguments");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_batch_rows_scatter_add: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (B == 0) return QREC_OK;
    QREC_REQUIRE(d_block && d_u && d_i && d_j && d_src, "qrec_batch_rows_scatter_add: null argument");
    const dim3 g(grid_for(3 * (int64_t)B * ld, 256)), b(256);
#define QREC_BS(L) hipLaunchKernelGGL((batch_rows_scatter_kernel<L>), g, b, 0, as_stream(stream), d_block, lo, hi, d_u, d_i, d_j, B, n_users, d_src)
    switch (ld) {
        case 32: QREC_BS(32); break;
        case 64: QREC_BS(64); break;
        case 128: QREC_BS(128); break;
        default: QREC_BS(256); break;
    }
#undef QREC_BS
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
