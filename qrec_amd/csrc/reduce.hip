// Full-table reductions used by the epoch-end loss terms (model/ranking/BPR.py:40:
// regU*(P*P).sum() + regI*(Q*Q).sum()).  Pure streaming read: 4 B/element, 16 B per lane per
// load.  The pad columns [d, ld) of a table are zero by contract (include/qrec_hip.h), so
// the whole [rows x ld] block is summed without any column test.
#include "common.h"

using namespace qrec;

namespace {

template <typename T, typename V4>
__global__ __launch_bounds__(256) void sumsq_kernel(const T *__restrict__ x, int64_t n_elems,
                                                    double *__restrict__ out) {
    double acc = 0.0;
    const int64_t n4 = n_elems >> 2;
    const V4 *x4 = reinterpret_cast<const V4 *>(x);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4;
         k += (int64_t)gridDim.x * blockDim.x) {
        const V4 v = x4[k];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_elems & 3)) {  // tail (ld is a multiple of 4 for fp32)
        const double v = (double)x[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
    __shared__ double s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// Epoch close of the numpy-path models, entirely on the device (model/ranking/BPR.py:40 + isConverged /
// updateLearningRate, base/iterativeRecommender.py:56-63,88-104):
//   loss = stats[0] + regU*sum(P*P) + regI*sum(Q*Q);  NaN/Inf -> failed (the reference exits)
//   converged = |lastLoss - loss| < tol;  if not: epoch > 1 -> lr *= 1.05 if |lastLoss| > |loss| else 0.5, cap at max_lr
// Every block reduces its share of both tables; the last block to arrive (ticket counter) takes the decision,
// appends the epoch to the log and clears the accumulators for the next epoch.
// The reference's decision at an epoch boundary (base/iterativeRecommender.py:56-63,88-104), from
// stats = {sum(-log sigma), sum P*P, sum Q*Q}: one thread.
__device__ inline void driver_decide(double *__restrict__ stats, double *__restrict__ state, double regU, double regI,
                                     double max_lr, double tol, double *__restrict__ log, int64_t log_capacity) {
    const double nll = atomicAdd(stats + 0, 0.0);     // the SGD kernels' f64 atomics live memory-side
    const double tp = stats[1], tq = stats[2];
    const double loss = nll + regU * tp + regI * tq;
    const double last = state[QREC_DRV_LAST_LOSS], lr_used = state[QREC_DRV_LR];
    const int64_t epoch = (int64_t)state[QREC_DRV_EPOCHS] + 1;
    double lr = lr_used;
    const bool finite = loss == loss && fabs(loss) <= 1.79769313486231570e308;
    const bool converged = finite && fabs(last - loss) < tol;
    if (finite && !converged) {
        if (epoch > 1) lr *= fabs(last) > fabs(loss) ? 1.05 : 0.5;
        if (max_lr > 0.0 && lr > max_lr) lr = max_lr;
    }
    state[QREC_DRV_LR] = lr;
    state[QREC_DRV_LAST_LOSS] = loss;
    state[QREC_DRV_EPOCHS] = (double)epoch;
    state[QREC_DRV_CONVERGED] = converged ? 1.0 : 0.0;
    state[QREC_DRV_FAILED] = finite ? 0.0 : 1.0;
    if (log && epoch <= log_capacity) {
        double *e = log + (epoch - 1) * QREC_DRV_LOG_WORDS;
        e[0] = loss; e[1] = lr_used; e[2] = nll; e[3] = last - loss; e[4] = tp; e[5] = tq;
    }
    stats[0] = 0.0;
}

template <typename T, typename V4>
__device__ inline double block_sumsq(const T *__restrict__ x, int64_t n_elems, double *s_part) {
    double acc = 0.0;
    const int64_t n4 = n_elems >> 2;
    const V4 *x4 = reinterpret_cast<const V4 *>(x);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        const V4 v = x4[k];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

template <typename T, typename V4>
__global__ __launch_bounds__(256) void epoch_close_kernel(const T *__restrict__ P, int64_t p_elems,
                                                          const T *__restrict__ Q, int64_t q_elems,
                                                          double *__restrict__ stats, double *__restrict__ state,
                                                          double regU, double regI, double max_lr, double tol,
                                                          double *__restrict__ log, int64_t log_capacity, int decide) {
    if (state && (state[QREC_DRV_CONVERGED] != 0.0 || state[QREC_DRV_FAILED] != 0.0)) return;
    __shared__ double s_part[4];
    __shared__ bool s_last;
    const double sp = block_sumsq<T, V4>(P, p_elems, s_part);
    const double sq = block_sumsq<T, V4>(Q, q_elems, s_part);
    // per-block partials in the stats buffer (plain stores, made visible by the fence), ONE same-address
    // atomic per block (the ticket); the last block adds the partials in block order: deterministic sums
    double *part = stats + QREC_STATS_PARTIALS;
    unsigned int *ticket = reinterpret_cast<unsigned int *>(stats + 3);
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + 2 * blockIdx.x, sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 2 * blockIdx.x + 1, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double tp = 0.0, tq = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
        tp += __hip_atomic_load(part + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tq += __hip_atomic_load(part + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { tp += __shfl_xor(tp, m, kWave); tq += __shfl_xor(tq, m, kWave); }
    __shared__ double s_tot[2][4];
    if ((threadIdx.x & 63) == 0) { s_tot[0][threadIdx.x >> 6] = tp; s_tot[1][threadIdx.x >> 6] = tq; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    tp = s_tot[0][0] + s_tot[0][1] + s_tot[0][2] + s_tot[0][3];
    tq = s_tot[1][0] + s_tot[1][1] + s_tot[1][2] + s_tot[1][3];
    if (!(decide & 2)) stats[1] = tp;
    if (!(decide & 4)) stats[2] = tq;
    *ticket = 0u;
    if (decide & 1) driver_decide(stats, state, regU, regI, max_lr, tol, log, log_capacity);
}

__global__ void epoch_decide_kernel(double *__restrict__ stats, double *__restrict__ state, double regU, double regI,
                                    double max_lr, double tol, double *__restrict__ log, int64_t log_capacity) {
    if (state[QREC_DRV_CONVERGED] != 0.0 || state[QREC_DRV_FAILED] != 0.0) return;
    driver_decide(stats, state, regU, regI, max_lr, tol, log, log_capacity);
}


// ---- multi-GPU epoch close, replicated item table (qrec_amd/dist.py) -------------------------------------------------
// Two launches around the step's one collective instead of five:
//   dist_pre :  delta = Q - Q_start  and  stats[1] = sum P*P              (P is this rank's user shard)
//   ... all-reduce {delta, stats[0..1]} ...
//   dist_post:  Q_start += delta; Q = Q_start;  stats[2] = sum Q*Q of the reconciled table;  the reference's decision.
// Block partials are added in block order by the last block (ticket), so the sums -- hence the decision -- are
// bit-identical on every rank.
__device__ inline bool last_block_total(double mine, int lane_of_pair, double *__restrict__ stats, double *total) {
    __shared__ bool s_last;
    __shared__ double s_tot[4];
    double *part = stats + QREC_STATS_PARTIALS;
    unsigned int *ticket = reinterpret_cast<unsigned int *>(stats + 3);
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + 2 * blockIdx.x + lane_of_pair, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return false;
    __threadfence();
    double t = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
        t += __hip_atomic_load(part + 2 * b + lane_of_pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, kWave);
    if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x != 0) return false;
    *total = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    *ticket = 0u;
    return true;
}

__global__ __launch_bounds__(256) void dist_pre_kernel(const float *__restrict__ P, int64_t p_elems, const float4 *__restrict__ Q,
                                                       const float4 *__restrict__ start, float4 *__restrict__ delta,
                                                       int64_t q4, double *__restrict__ stats, const double *__restrict__ state) {
    if (state && (state[QREC_DRV_CONVERGED] != 0.0 || state[QREC_DRV_FAILED] != 0.0)) return;
    __shared__ double s_part[4];
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < q4; k += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = Q[k], b = start[k];
        delta[k] = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
    const double sp = block_sumsq<float, float4>(P, p_elems, s_part);
    double total;
    if (last_block_total(sp, 0, stats, &total)) stats[1] = total;
}

__global__ __launch_bounds__(256) void dist_post_kernel(float4 *__restrict__ Q, float4 *__restrict__ start,
                                                        const float4 *__restrict__ delta, int64_t q4, double *__restrict__ stats,
                                                        double *__restrict__ state, double regU, double regI, double max_lr,
                                                        double tol, double *__restrict__ log, int64_t log_capacity) {
    if (state[QREC_DRV_CONVERGED] != 0.0 || state[QREC_DRV_FAILED] != 0.0) return;
    double acc = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < q4; k += (int64_t)gridDim.x * blockDim.x) {
        const float4 s = start[k], d = delta[k];
        const float4 r = make_float4(s.x + d.x, s.y + d.y, s.z + d.z, s.w + d.w);
        start[k] = r;
        Q[k] = r;
        acc += (double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z + (double)r.w * r.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
    __shared__ double s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    const double sq = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    double total;
    if (last_block_total(sq, 1, stats, &total)) {
        stats[2] = total;
        driver_decide(stats, state, regU, regI, max_lr, tol, log, log_capacity);
    }
}

}  // namespace

namespace {
int launch_epoch_close(const void *d_P, int64_t p_rows, const void *d_Q, int64_t q_rows, int dtype, int32_t ld,
                       double *d_stats, double *d_state, double regU, double regI, double max_lr, double tol,
                       double *d_log, int64_t log_capacity, int decide, hipStream_t st) {
    const int64_t pe = p_rows * (int64_t)ld, qe = q_rows * (int64_t)ld;
    int64_t blocks = ((pe > qe ? pe : qe) / 4 + 255) / 256;
    if (blocks > QREC_STATS_MAX_BLOCKS) blocks = QREC_STATS_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    if (dtype == QREC_F32)
        hipLaunchKernelGGL((epoch_close_kernel<float, float4>), dim3((unsigned)blocks), dim3(256), 0, st, (const float *)d_P, pe,
                           (const float *)d_Q, qe, d_stats, d_state, regU, regI, max_lr, tol, d_log, log_capacity, decide);
    else
        hipLaunchKernelGGL((epoch_close_kernel<double, double4>), dim3((unsigned)blocks), dim3(256), 0, st, (const double *)d_P, pe,
                           (const double *)d_Q, qe, d_stats, d_state, regU, regI, max_lr, tol, d_log, log_capacity, decide);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}
}  // namespace

extern "C" int qrec_epoch_close(const void *d_P, int64_t p_rows, const void *d_Q, int64_t q_rows, int dtype, int32_t ld,
                                double *d_stats, double *d_state, double regU, double regI, double max_lr,
                                double tol, double *d_log, int64_t log_capacity, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_stats && d_state && p_rows >= 0 && q_rows >= 0 && ld >= 4 && ld % 4 == 0,
                 "qrec_epoch_close: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_epoch_close: bad dtype %d", dtype);
    QREC_REQUIRE(log_capacity == 0 || d_log, "qrec_epoch_close: log capacity without a log");
    return launch_epoch_close(d_P, p_rows, d_Q, q_rows, dtype, ld, d_stats, d_state, regU, regI, max_lr, tol, d_log,
                              log_capacity, 1, as_stream(stream));
}

// The two halves of qrec_epoch_close for runs whose loss terms are summed over ranks in between (multi-GPU:
// all-reduce d_stats[0..1] = {sum(-log sigma), sum P*P} after the sums, Q is replicated).
extern "C" int qrec_epoch_sums(const void *d_P, int64_t p_rows, const void *d_Q, int64_t q_rows, int dtype, int32_t ld,
                               double *d_stats, const double *d_state, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_stats && p_rows >= 0 && q_rows >= 0 && ld >= 4 && ld % 4 == 0, "qrec_epoch_sums: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_epoch_sums: bad dtype %d", dtype);
    return launch_epoch_close(d_P, p_rows, d_Q, q_rows, dtype, ld, d_stats, const_cast<double *>(d_state), 0, 0, 0, 0,
                              nullptr, 0, 0, as_stream(stream));
}
// One table's deterministic sum of squares into d_stats[slot] (slot 1 = users, 2 = items), leaving the other slot alone:
// lets a multi-GPU step take sum P*P before its one fused collective and sum Q*Q after the item table is reconciled.
extern "C" int qrec_epoch_sum_table(const void *d_X, int64_t rows, int dtype, int32_t ld, double *d_stats, int slot,
                                    const double *d_state, void *stream) {
    QREC_REQUIRE(d_X && d_stats && rows >= 0 && ld >= 4 && ld % 4 == 0, "qrec_epoch_sum_table: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_epoch_sum_table: bad dtype %d", dtype);
    QREC_REQUIRE(slot == 1 || slot == 2, "qrec_epoch_sum_table: slot must be 1 (sum P*P) or 2 (sum Q*Q)");
    return slot == 1 ? launch_epoch_close(d_X, rows, d_X, 0, dtype, ld, d_stats, const_cast<double *>(d_state), 0, 0, 0, 0, nullptr,
                                          0, 4, as_stream(stream))
                     : launch_epoch_close(d_X, 0, d_X, rows, dtype, ld, d_stats, const_cast<double *>(d_state), 0, 0, 0, 0, nullptr,
                                          0, 2, as_stream(stream));
}
extern "C" int qrec_epoch_decide(double *d_stats, double *d_state, double regU, double regI, double max_lr, double tol,
                                 double *d_log, int64_t log_capacity, void *stream) {
    QREC_REQUIRE(d_stats && d_state, "qrec_epoch_decide: null argument");
    QREC_REQUIRE(log_capacity == 0 || d_log, "qrec_epoch_decide: log capacity without a log");
    hipLaunchKernelGGL(epoch_decide_kernel, dim3(1), dim3(1), 0, as_stream(stream), d_stats, d_state, regU, regI, max_lr, tol,
                       d_log, log_capacity);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

extern "C" int qrec_sumsq(const void *d_x, int dtype, int64_t rows, int32_t d, int32_t ld,
                          double *d_out, void *stream) {
    QREC_REQUIRE(d_out && rows >= 0 && d >= 1 && ld >= d, "qrec_sumsq: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_sumsq: bad dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    QREC_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(double), st));
    if (rows == 0) return QREC_OK;
    QREC_REQUIRE(d_x, "qrec_sumsq: null table");
    const int64_t n = rows * (int64_t)ld;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (dtype == QREC_F32)
        hipLaunchKernelGGL((sumsq_kernel<float, float4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float *)d_x, n, d_out);
    else
        hipLaunchKernelGGL((sumsq_kernel<double, double4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const double *)d_x, n, d_out);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

extern "C" int qrec_dist_epoch_pre(const float *d_P, int64_t p_rows, int32_t ld, const float *d_Q, const float *d_Q_start,
                                   float *d_delta, int64_t q_rows, double *d_stats, const double *d_state, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_Q_start && d_delta && d_stats && p_rows >= 0 && q_rows >= 0 && ld >= 4 && ld % 4 == 0,
                 "qrec_dist_epoch_pre: bad arguments");
    const int64_t pe = p_rows * (int64_t)ld, q4 = q_rows * (int64_t)ld / 4;
    int64_t blocks = ((pe / 4 > q4 ? pe / 4 : q4) + 255) / 256;
    if (blocks > QREC_STATS_MAX_BLOCKS) blocks = QREC_STATS_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dist_pre_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_P, pe, (const float4 *)d_Q,
                       (const float4 *)d_Q_start, (float4 *)d_delta, q4, d_stats, d_state);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

extern "C" int qrec_dist_epoch_post(float *d_Q, float *d_Q_start, const float *d_delta, int64_t q_rows, int32_t ld,
                                    double *d_stats, double *d_state, double regU, double regI, double max_lr, double tol,
                                    double *d_log, int64_t log_capacity, void *stream) {
    QREC_REQUIRE(d_Q && d_Q_start && d_delta && d_stats && d_state && q_rows >= 0 && ld >= 4 && ld % 4 == 0,
                 "qrec_dist_epoch_post: bad arguments");
    QREC_REQUIRE(log_capacity == 0 || d_log, "qrec_dist_epoch_post: log capacity without a log");
    const int64_t q4 = q_rows * (int64_t)ld / 4;
    int64_t blocks = (q4 + 255) / 256;
    if (blocks > QREC_STATS_MAX_BLOCKS) blocks = QREC_STATS_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dist_post_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (float4 *)d_Q, (float4 *)d_Q_start,
                       (const float4 *)d_delta, q4, d_stats, d_state, regU, regI, max_lr, tol, d_log, log_capacity);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}
