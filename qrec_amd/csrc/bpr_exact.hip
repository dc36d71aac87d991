// Order-exact BPR beyond one wavefront: the static schedule of exact_schedule.cpp executed by ONE workgroup.
//
// model/ranking/BPR.py:45-53 applied strictly in the reference's order is a dependence DAG ~n/6 deep (every user run is a
// chain through P[u], popular items link the runs).  The walker of bpr_sgd.hip (one wavefront, one triplet after the
// other, ~1 us each) leaves the DAG's width unused.  Here a workgroup of NW wavefronts advances one SCHEDULE STEP per
// barrier: wavefront w applies the step's w-th triplet -- all triplets of a step are independent, every dependence points
// at least one step back.  What bounds a step is the latency of one triplet's dependent chain, so everything else is
// taken off it:
//   * rows rewritten one or two steps ago are forwarded through LDS (3 rotating buffers of NW x 3 rows), which is where
//     every user run's P[u] lives from triplet to triplet;
//   * rows last written three or more steps ago are loaded from the table TWO steps ahead of their use, into registers
//     (the schedule is known, so the loads are issued under the preceding steps' arithmetic);
//   * new rows go to the table with plain stores that nobody waits for: the workgroup's wavefronts share one CU and one
//     L1, stores issued before a barrier are observed by loads issued after it (LLVM AMDGPU memory model, workgroup
//     scope, non-tgsplit mode), and nothing outside the workgroup reads the tables during the launch;
//   * -log(sigmoid(x)) is not on the chain: x is logged per triplet and summed by a second, parallel kernel.
// The arithmetic of a triplet is bpr_ordered_kernel's statement for statement (no contraction, same cross-lane sum),
// so both kernels produce the same bits for the same order.
#include <cmath>

#include "common.h"

using namespace qrec;

namespace {

template <typename T> __device__ inline T dev_exp(T x);
template <> __device__ inline float dev_exp<float>(float x) { return expf(x); }
template <> __device__ inline double dev_exp<double>(double x) { return exp(x); }

__device__ inline double neg_log_sigmoid_d(double x) { return x >= 0.0 ? log1p(exp(-x)) : (-x + log1p(exp(x))); }

template <typename T>
__device__ inline T wave_allreduce_sum(T v) { return wave_sum_dpp(v); }   // common.h: DPP, not the LDS crossbar

struct Entry {      // as loaded: nothing is computed from a/b before the step that uses them (a use is a wait)
    int4 a, b;      // {u, i, j, t}, {src_P, src_Qi, src_Qj, -}
    bool valid;
    __device__ int src(int k) const { return valid ? (k == 0 ? b.x : k == 1 ? b.y : b.z) : 0; }   // invalid: "forwarded", no table load
};

// Every wavefront executes the same straight-line instruction stream at every step, whether it has a triplet or not: a
// wavefront without one (and every lane past d) computes on a dummy row and stores to it.  That is what lets the
// compiler count its outstanding memory operations exactly -- the rows of step s+2 and the schedule entry of step s+4
// stay in flight across the barriers, waited for with s_waitcnt vmcnt(N > 0) right where they are used, instead of
// vmcnt(0) at every control-flow merge.  Register sets are renamed by unrolling (entries: 4 sets, rows: 2), never copied:
// a copy would be a use.
template <typename T, int EPL>
__global__ __launch_bounds__(1024) void bpr_levels_kernel(T *__restrict__ P, T *__restrict__ Q, int d, int ld,
                                                          const int4 *__restrict__ entries, const int32_t *__restrict__ step_off,
                                                          int n_steps, int nw, T lr, T cu, T ci, T *__restrict__ xlog,
                                                          T *__restrict__ dummy) {
#pragma clang fp contract(off)  // numpy rounds every product and sum separately
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROW = 64 * EPL;
    T *fwd = reinterpret_cast<T *>(smem);                      // [3 steps][nw slots][3 rows][ROW]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T *const my_dummy = dummy + (int64_t)w * ROW + lane;       // this lane's column of the wavefront's dummy row
    T *const x_dummy = dummy + (int64_t)QREC_EXACT_MAX_WIDTH * ROW + lane;
    bool col_ok[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) col_ok[e] = (lane + 64 * e) < d;

    auto load_entry = [&](int s) {
        // steps past the end and slots past the step's width read entry 0 and are marked invalid (no branch)
        const int sc = s < n_steps ? s : 0;
        const int o = step_off[sc], width = step_off[sc + 1] - o;
        const bool ok = s < n_steps && w < width;
        const int64_t at = ok ? (int64_t)(o + w) : 0;
        Entry en;
        en.a = entries[2 * at]; en.b = entries[2 * at + 1];
        en.valid = ok;
        return en;
    };
    auto row_ptr = [&](const Entry &en, int k, int e) -> T * {
        T *tab = k == 0 ? P + (int64_t)en.a.x * ld : Q + (int64_t)(k == 1 ? en.a.y : en.a.z) * ld;
        return (en.valid && col_ok[e]) ? tab + lane + 64 * e : my_dummy + 64 * e;
    };
    // rows the table still holds current (last toucher >= 3 steps back), fetched two steps ahead of their use
    auto load_rows = [&](const Entry &en, T (&r)[3][EPL]) {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const T *p = en.src(k) < 0 ? row_ptr(en, k, e) : my_dummy + 64 * e;
                r[k][e] = *p;
            }
    };

    Entry E[4];
    T R[2][3][EPL];
    // prologue: the same sequence of memory operations a steady-state step leaves in flight (entry, rows, entry, 3*EPL+1
    // stores, rows, entry), so that the waits at the loop head are the steady-state ones and not a drain
    auto dummy_stores = [&]() {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int e = 0; e < EPL; e++) my_dummy[64 * e] = T(0);
        *x_dummy = T(0);
    };
    E[0] = load_entry(0);
    E[1] = load_entry(1);
    load_rows(E[0], R[0]);
    E[2] = load_entry(2);
    dummy_stores();
    load_rows(E[1], R[1]);
    E[3] = load_entry(3);

    auto step = [&](int s, Entry &en, T (&r)[3][EPL], const Entry &ahead2) {
        // the schedule entry of step s+4 first: it takes this entry's registers at the end of the step, and by then --
        // a whole step later -- the copy no longer has to wait for it (issued last it would drain the row loads too)
        const Entry later = load_entry(s + 4);
        T row[3][EPL] = {};
        T x = 0;
        if (en.valid) {      // an idle wavefront skips the arithmetic (not the memory operations: their count stays fixed)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int sk = en.src(k), code = sk < 0 ? 0 : sk;
                // forwarded: written `dist` steps ago by slot `slot` as its row `which`
                const int which = code % 3, slot = (code / 3) % QREC_EXACT_MAX_WIDTH, dist = code / (3 * QREC_EXACT_MAX_WIDTH) + 1;
                const T *f = fwd + (((int64_t)((s + 3 - dist) % 3) * nw + slot) * 3 + which) * ROW + lane;
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const T v = f[64 * e];
                    row[k][e] = sk < 0 ? (col_ok[e] ? r[k][e] : T(0)) : v;     // lanes past d read the dummy row: they hold 0
                }
            }
            T di = 0, dj = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) { di += row[0][e] * row[1][e]; dj += row[0][e] * row[2][e]; }
            di = wave_allreduce_sum(di); dj = wave_allreduce_sum(dj);
            x = di - dj;
            const T sg = T(1) / (T(1) + dev_exp<T>(-x));
            const T g = lr * (T(1) - sg);
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                row[0][e] += g * (row[1][e] - row[2][e]);
                row[1][e] += g * row[0][e];
                row[2][e] -= g * row[0][e];
                row[0][e] -= cu * row[0][e];
                row[1][e] -= ci * row[1][e];
                row[2][e] -= ci * row[2][e];
            }
        }
        T *f = fwd + ((int64_t)(s % 3) * nw + w) * 3 * ROW + lane;
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                f[k * ROW + 64 * e] = row[k][e];
                *row_ptr(en, k, e) = row[k][e];
            }
        *((en.valid && lane == 0) ? xlog + en.a.w : x_dummy) = x;
        load_rows(ahead2, r);               // step s+2's table rows: their last touchers are <= s-1, stored before the last barrier
        en = later;
        __syncthreads();
    };

    for (int s = 0; s < n_steps; s += 4) {      // n_steps is padded to a multiple of 4 by the caller's schedule (empty steps)
        step(s, E[0], R[0], E[2]);
        step(s + 1, E[1], R[1], E[3]);
        step(s + 2, E[2], R[0], E[0]);
        step(s + 3, E[3], R[1], E[1]);
    }
}

// sum over the epoch's triplets of -log(sigmoid(x)): model/ranking/BPR.py:53 from the logged x.  fp64 tables: the
// reference's own expression -log(1/(1+exp(-x))); fp32: the stable form (bpr_sgd.hip).  Block partials are added in
// block order by the last block: a deterministic sum.
template <typename T>
__global__ __launch_bounds__(256) void nll_sum_kernel(const T *__restrict__ xlog, int64_t n, double *__restrict__ partial,
                                                      unsigned int *__restrict__ ticket, double *__restrict__ out) {
    double acc = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)xlog[t];
        if constexpr (sizeof(T) == 8) acc += -log(1.0 / (1.0 + exp(-x)));
        else acc += neg_log_sigmoid_d(x);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
    __shared__ double s_part[4];
    __shared__ bool s_last;
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, s_part[0] + s_part[1] + s_part[2] + s_part[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    double tot = 0.0;
    for (unsigned b = 0; b < gridDim.x; b++) tot += __hip_atomic_load(partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out = tot;
    *ticket = 0u;
}

constexpr int kNllBlocks = 128;

template <typename T, int EPL>
int launch_levels(void *P, void *Q, int d, int ld, const int32_t *entries, const int32_t *off, int n_steps, int nw, double lr,
                  double regU, double regI, void *xlog, int64_t n, hipStream_t st) {
    const T tlr = (T)lr, cu = (T)lr * (T)regU, ci = (T)lr * (T)regI;  // numpy: (lr*reg)*row
    const size_t lds = (size_t)3 * nw * 3 * 64 * EPL * sizeof(T);
    static size_t configured = 0;
    if (lds > configured) {
        QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&bpr_levels_kernel<T, EPL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    hipLaunchKernelGGL((bpr_levels_kernel<T, EPL>), dim3(1), dim3(64 * nw), lds, st, (T *)P, (T *)Q, d, ld, (const int4 *)entries, off,
                       n_steps, nw, tlr, cu, ci, (T *)xlog, (T *)xlog + n);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // namespace

extern "C" {

int qrec_bpr_exact_width(int dtype, int32_t d, int32_t *width) {
    QREC_REQUIRE(width && d >= 1 && d <= 256 && (dtype == QREC_F32 || dtype == QREC_F64), "qrec_bpr_exact_width: bad arguments");
    const int epl = d <= 64 ? 1 : d <= 128 ? 2 : 4;
    const size_t per_wave = (size_t)3 * 3 * 64 * epl * (dtype == QREC_F64 ? 8 : 4);
    int nw = (int)((size_t)144 * 1024 / per_wave);              // of the CU's 160 KiB
    if (nw > QREC_EXACT_MAX_WIDTH) nw = QREC_EXACT_MAX_WIDTH;
    *width = nw;
    return QREC_OK;
}

int qrec_bpr_sgd_scheduled(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld, const int32_t *d_entries,
                           const int32_t *d_step_off, int64_t n_steps, int32_t width, int64_t n, double lr, double regU,
                           double regI, void *d_xlog, double *d_scratch, double *d_loss, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_loss && n >= 0 && n_steps >= 0 && n_steps < (1ll << 31), "qrec_bpr_sgd_scheduled: bad arguments");
    QREC_REQUIRE(d >= 1 && d <= 256 && ld >= d, "qrec_bpr_sgd_scheduled: need 1 <= d <= 256, ld >= d (got d=%d ld=%d)", d, ld);
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_bpr_sgd_scheduled: bad dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    if (n == 0) { QREC_HIP_CHECK(hipMemsetAsync(d_loss, 0, sizeof(double), st)); return QREC_OK; }
    QREC_REQUIRE(d_entries && d_step_off && d_xlog && d_scratch, "qrec_bpr_sgd_scheduled: null schedule / scratch");
    int32_t max_w = 0;
    qrec_bpr_exact_width(dtype, d, &max_w);
    QREC_REQUIRE(width >= 1 && width <= max_w, "qrec_bpr_sgd_scheduled: width %d outside 1..%d for this table", width, max_w);
    int rc;
#define QREC_LV(T, EPL) rc = launch_levels<T, EPL>(d_P, d_Q, d, ld, d_entries, d_step_off, (int)n_steps, width, lr, regU, regI, d_xlog, n, st)
    if (dtype == QREC_F64) { if (d <= 64) QREC_LV(double, 1); else if (d <= 128) QREC_LV(double, 2); else QREC_LV(double, 4); }
    else { if (d <= 64) QREC_LV(float, 1); else if (d <= 128) QREC_LV(float, 2); else QREC_LV(float, 4); }
#undef QREC_LV
    if (rc != QREC_OK) return rc;
    // d_scratch: kNllBlocks partials + the ticket word (zero before the first use; the kernel re-arms it)
    unsigned int *ticket = reinterpret_cast<unsigned int *>(d_scratch + kNllBlocks);
    if (dtype == QREC_F64)
        hipLaunchKernelGGL((nll_sum_kernel<double>), dim3(kNllBlocks), dim3(256), 0, st, (const double *)d_xlog, n, d_scratch, ticket, d_loss);
    else
        hipLaunchKernelGGL((nll_sum_kernel<float>), dim3(kNllBlocks), dim3(256), 0, st, (const float *)d_xlog, n, d_scratch, ticket, d_loss);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
