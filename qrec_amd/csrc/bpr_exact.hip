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
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

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
// (launch bound: 16 wavefronts, except rows of 256 fp64 elements -- EPL = 4 doubles per lane: qrec_bpr_exact_width caps those at 8
// wavefronts by their LDS footprint anyway, and under the 128-VGPR budget of a 1024-thread block that instantiation spilled 24
// VGPRs / 84 B of scratch per lane; a 512-thread bound gives it 256)
template <typename T, int EPL>
__global__ __launch_bounds__((sizeof(T) * EPL >= 32) ? 512 : 1024) void bpr_levels_kernel(T *__restrict__ P, T *__restrict__ Q, int d, int ld,
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
                const int which = code & 3, slot = (code >> 2) % QREC_EXACT_MAX_WIDTH, dist = code / (4 * QREC_EXACT_MAX_WIDTH) + 1;
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

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: FOUR triplets per wavefront, rows handed on in REGISTERS or through the table, nothing in LDS.
//
// A triplet's rows live on the 16 lanes of one DPP row (lane l of the group holds the EPG = ld/16 contiguous elements
// l*EPG ...), so a wavefront advances four independent triplets of the step with ONE instruction stream; a step of width 8
// needs two wavefronts.  How it got here, measured at the Yelp2018 shape, fp64, width 8 (profiles/r03_exact_mode_probe.json,
// profiles/r03_exact_step_breakdown.json):
//   one triplet per wavefront (the kernel above)                                  0.92 us/step  214 k steps   6.3 M triplets/s
//   four per wavefront, rows forwarded through LDS buffers as above               0.70          214 k         8.3
//   ... and a user's run kept on ONE slot, P[u] in that group's registers         0.365         229 k        15.0   (this kernel)
// What each step bought:
//   * a one-triplet wavefront is bound by the ISSUE of its ~200 dependent VALU instructions (47 % busy, 53 % waiting on itself);
//     four triplets share one stream of ~130;
//   * the cross-lane sum is 4 DPP rotations inside the row (row_ror 8, 4, 2, 1: every lane ends with the same bits) instead of
//     5 row steps + 2 row broadcasts + a readlane, and ONE dot product P[u].(Q[i] - Q[j]) instead of two -- the difference of
//     the two sums the reference forms (BPR.py:46) regrouped; 1e-16-class, the parity budget of the mode is 1e-10;
//   * exp and the division are written out (Cody-Waite reduction + degree-13 Taylor in Estrin form, v_rcp_f64 + two Newton
//     steps): ~25 instructions with a dependent depth of ~12 where the library calls have ~45 in one chain;
//   * of the LDS variant's 0.70 us, 0.14 was the step-to-step dependence through the forwarding buffers (ds_write, the wait
//     for it, the barrier, ds_read and its latency -- on the chain of 86 % of the steps, because a user's run hands P[u] on
//     every step), 0.09 two dependent scalar loads of step_off (they share lgkmcnt with the LDS traffic), and 24 VALU
//     instructions were selects between the LDS copy and the prefetched copy of a row.  Here (schedule:
//     qrec_bpr_exact_schedule_reg) a run stays on ONE slot and P[u] simply stays in that group's registers; every other row
//     comes from the table, which the schedule makes current (last toucher >= 3 steps back, loads issued two steps ahead):
//     6.6 % more steps, each half as long.  The schedule is read in a fixed-width layout (qrec_bpr_exact_expand).
// Same memory-operation discipline as the kernel above -- every group issues the same loads and stores every step -- with
// three lessons from the ISA of this one: (1) "nothing to load / store" is an offset past the end of a buffer descriptor
// (hardware bounds check: load 0 / store nothing, no traffic, no branch; exec-masked `if`s around stores made the compiler
// lose count of the operations in flight); (2) no prologue: the layout starts with four empty steps and the register sets
// start empty, so the waits at the loop head are the steady state's own (the compiler reorders a hand-written prologue's
// independent loads and stores); (3) a scheduling barrier after each step's s_barrier, or the next step's arithmetic on
// rows still in flight is hoisted up and drags their wait along.
// Arithmetic differs from the one-wavefront walker's in the last bits (other summation tree, other exp): held to each other
// at 1e-13, and this kernel to ITSELF bit for bit across widths (any schedule of the same order gives the same bits).
template <typename T> struct Fast;
template <> struct Fast<double> {
    static __device__ inline double exp_(double y) {        // e^y, |rel err| ~ 3e-16; y clamped to [-700, 700]
        y = __builtin_fmin(__builtin_fmax(y, -700.0), 700.0);
        const double k = __builtin_rint(y * 1.44269504088896338700e+00);
        double r = __builtin_fma(-k, 6.93147180369123816490e-01, y);          // ln2 = hi + lo (fdlibm split: k * hi is exact)
        r = __builtin_fma(-k, 1.90821492927058770002e-10, r);                 // |r| <= 0.3466
        const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
        const double a0 = __builtin_fma(1.0, r, 1.0);
        const double a1 = __builtin_fma(1.0 / 6, r, 0.5);
        const double a2 = __builtin_fma(1.0 / 120, r, 1.0 / 24);
        const double a3 = __builtin_fma(1.0 / 5040, r, 1.0 / 720);
        const double a4 = __builtin_fma(1.0 / 362880, r, 1.0 / 40320);
        const double a5 = __builtin_fma(1.0 / 39916800, r, 1.0 / 3628800);
        const double a6 = __builtin_fma(1.0 / 6227020800.0, r, 1.0 / 479001600);
        const double b0 = __builtin_fma(a1, r2, a0), b1 = __builtin_fma(a3, r2, a2), b2 = __builtin_fma(a5, r2, a4);
        const double c0 = __builtin_fma(b1, r4, b0), c1 = __builtin_fma(a6, r4, b2);
        return __builtin_ldexp(__builtin_fma(c1, r8, c0), (int)k);
    }
    static __device__ inline double rcp_(double a) {        // 1/a for a in [1, 1e305]: v_rcp_f64 (~2^-23) + two Newton steps
        double r = __builtin_amdgcn_rcp(a);
        r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
        return r;
    }
};
template <> struct Fast<float> {
    // y clamped to <= 88: e^88 = 1.65e38 is finite in fp32, so 1 + e^y stays finite and rcp_'s Newton step never sees
    // inf * 0 (x = -120: sigma is 0 either way -- tests/test_gpu_bpr.py::test_loss_is_finite_where_fp32_sigmoid_underflows)
    static __device__ inline float exp_(float y) { return expf(__builtin_fminf(y, 88.0f)); }
    static __device__ inline float rcp_(float a) {
        float r = __builtin_amdgcn_rcpf(a);
        return __builtin_fmaf(__builtin_fmaf(-a, r, 1.f), r, r);
    }
};

template <typename T>
__device__ inline T group_sum16(T v) {       // the sum over the 16 lanes of a DPP row, in every lane of the row, identical bits
    v = v + dpp_take<0x128, 0xf, 0xf>(v);    // row_ror:8
    v = v + dpp_take<0x124, 0xf, 0xf>(v);    // row_ror:4
    v = v + dpp_take<0x122, 0xf, 0xf>(v);    // row_ror:2
    v = v + dpp_take<0x121, 0xf, 0xf>(v);    // row_ror:1
    return v;
}

template <typename T, int EPG> struct alignas((sizeof(T) * EPG >= 16) ? 16 : sizeof(T) * EPG) RowVec { T v[EPG]; };

template <typename T, int EPG>
__device__ inline RowVec<T, EPG> buf_load_row(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    constexpr int NB = (int)sizeof(T) * EPG;
    RowVec<T, EPG> v;
    if constexpr (NB >= 16) {
        u32x4 w[NB / 16];
#pragma unroll
        for (int c = 0; c < NB / 16; c++) w[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u * c), 0, 0);
        __builtin_memcpy(&v, w, NB);
    } else if constexpr (NB == 8) {
        const auto w = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
        __builtin_memcpy(&v, &w, 8);
    } else {
        const auto w = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
        __builtin_memcpy(&v, &w, 4);
    }
    return v;
}
template <typename T, int EPG>
__device__ inline void buf_store_row(__amdgpu_buffer_rsrc_t rs, uint32_t off, const RowVec<T, EPG> &v) {
    constexpr int NB = (int)sizeof(T) * EPG;
    if constexpr (NB >= 16) {
        u32x4 w[NB / 16];
        __builtin_memcpy(w, &v, NB);
#pragma unroll
        for (int c = 0; c < NB / 16; c++) __builtin_amdgcn_raw_buffer_store_b128(w[c], rs, (int)(off + 16u * c), 0, 0);
    } else if constexpr (NB == 8) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 w; __builtin_memcpy(&w, &v, 8);
        __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)off, 0, 0);
    } else {
        uint32_t w; __builtin_memcpy(&w, &v, 4);
        __builtin_amdgcn_raw_buffer_store_b32(w, rs, (int)off, 0, 0);
    }
}

template <typename T, int EPG, int DBG = 0>
__global__ __launch_bounds__(256) void bpr_levels_reg_kernel(T *__restrict__ P, T *__restrict__ Q, uint32_t p_bytes, uint32_t q_bytes,
                                                             const int4 *__restrict__ wide, int n_steps, T lr, T cu, T ci,
                                                             T *__restrict__ xlog, uint32_t x_bytes) {
    constexpr int ROW = 16 * EPG;                              // == ld
    constexpr uint32_t kNowhere = 0xFFFFFF00u;                 // past the end of every descriptor: load 0 / store nothing
    using Vec = RowVec<T, EPG>;
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    const int slot = (int)(threadIdx.x >> 6) * 4 + (lane >> 4), slots = (int)(blockDim.x >> 6) * 4;
    const uint32_t col_b = (uint32_t)(l16 * EPG) * (uint32_t)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsP = make_rsrc(P, p_bytes), rsQ = make_rsrc(Q, q_bytes), rsX = make_rsrc(xlog, x_bytes);
    struct WEntry { int4 a, b; };     // {u or -1, i, j, t}, {src_P (-2 registers, -1 table), -, -, bit 0: P goes on in registers}
    auto load_entry = [&](int s) {
        WEntry en;
        const int64_t at = (int64_t)s * slots + slot;
        en.a = wide[2 * at]; en.b = wide[2 * at + 1];
        return en;
    };
    auto row_off = [&](int row) { return (uint32_t)row * (uint32_t)(ROW * sizeof(T)) + col_b; };
    auto load_rows = [&](const WEntry &en, Vec (&r)[3]) {
        const bool valid = (en.a.x >= 0) & !(DBG & 2);          // & not &&: no control flow, the memory operations stay countable
        r[0] = buf_load_row<T, EPG>(rsP, (valid & (en.b.x != -2)) ? row_off(en.a.x) : kNowhere);
        r[1] = buf_load_row<T, EPG>(rsQ, valid ? row_off(en.a.y) : kNowhere);
        r[2] = buf_load_row<T, EPG>(rsQ, valid ? row_off(en.a.z) : kNowhere);
    };
    // No prologue: the wide layout starts with QREC_EXACT_WIDE_LEAD (4) empty steps, the register sets start as "no triplet" /
    // zeros WITHOUT a load, and the loop begins at the first of those steps.  Nothing is in flight when the loop is entered, so
    // the waits the compiler places inside are the steady state's own (a hand-written prologue has to reproduce the exact
    // sequence of operations a step leaves in flight -- and the compiler reorders independent loads and stores).
    WEntry E[4];
    Vec R[2][3];
    Vec carry;                                                  // the P row this group wrote in the previous step
#pragma unroll
    for (int q = 0; q < 4; q++) { E[q].a = make_int4(-1, 0, 0, 0); E[q].b = make_int4(-1, -1, -1, 0); }
#pragma unroll
    for (int e = 0; e < EPG; e++) {
        carry.v[e] = T(0);
#pragma unroll
        for (int q = 0; q < 2; q++) { R[q][0].v[e] = T(0); R[q][1].v[e] = T(0); R[q][2].v[e] = T(0); }
    }

    auto step = [&](int s, WEntry &en, Vec (&r)[3], const WEntry &ahead2) {
        const bool valid = (en.a.x >= 0) & !(DBG & 1), from_reg = en.b.x == -2;
        T p[EPG], diff[EPG];
        T acc = T(0);
#pragma unroll
        for (int e = 0; e < EPG; e++) {
            p[e] = from_reg ? carry.v[e] : r[0].v[e];
            diff[e] = r[1].v[e] - r[2].v[e];
            acc = __builtin_fma(p[e], diff[e], acc);
        }
        const T x = group_sum16(acc);                                         // P[u].Q[i] - P[u].Q[j], BPR.py:46
        const T sg = (DBG & 4) ? x * T(0.001) : Fast<T>::rcp_(T(1) + Fast<T>::exp_(-x));   // util/qmath.py:127-128
        const T g = lr * (T(1) - sg);
        Vec out[3];
#pragma unroll
        for (int e = 0; e < EPG; e++) {                                       // BPR.py:47-52, statement for statement
            T pp = p[e], qi = r[1].v[e], qj = r[2].v[e];
            pp += g * diff[e];
            qi += g * pp;
            qj -= g * pp;
            pp -= cu * pp;
            qi -= ci * qi;
            qj -= ci * qj;
            out[0].v[e] = pp; out[1].v[e] = qi; out[2].v[e] = qj;
        }
        carry = out[0];
        buf_store_row<T, EPG>(rsP, (valid & !(en.b.w & 1)) ? row_off(en.a.x) : kNowhere, out[0]);
        buf_store_row<T, EPG>(rsQ, valid ? row_off(en.a.y) : kNowhere, out[1]);
        buf_store_row<T, EPG>(rsQ, valid ? row_off(en.a.z) : kNowhere, out[2]);
        RowVec<T, 1> xv; xv.v[0] = x;
        buf_store_row<T, 1>(rsX, (valid & (l16 == 0)) ? (uint32_t)en.a.w * (uint32_t)sizeof(T) : kNowhere, xv);
        load_rows(ahead2, r);
        en = load_entry(s + 4);      // straight into this set's registers (its last use was the stores above): no copy, hence no wait;
                                     // first needed two steps on, for the row offsets of step s + 4
        if constexpr (!(DBG & 8)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);      // nothing of the next step moves up here: its arithmetic on rows that are still in
                                                // flight would pull their wait into this step (seen in the ISA: vmcnt(6))
    };
    for (int s = 0; s < n_steps; s += 4) {
        step(s, E[0], R[0], E[2]);
        step(s + 1, E[1], R[1], E[3]);
        step(s + 2, E[2], R[0], E[0]);
        step(s + 3, E[3], R[1], E[1]);
    }
}

// CSR schedule -> wide layout: [n_steps + QREC_EXACT_WIDE_PAD steps][slots] entries of 8 ints, empty slots u = -1 (the buffer is
// filled with 0xFF first)
__global__ __launch_bounds__(256) void expand_schedule_kernel(const int4 *__restrict__ entries, const int32_t *__restrict__ step_off,
                                                              int n_steps, int slots, int lead, int4 *__restrict__ wide) {
    for (int s = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4); s < n_steps; s += gridDim.x * (blockDim.x / 16)) {
        const int o = step_off[s], w = step_off[s + 1] - o, k = threadIdx.x & 15;
        if (k < w) {
            const int64_t from = (int64_t)(o + k) * 2;
            const int4 a = entries[from], b = entries[from + 1];
            const int to_slot = (b.w & 0x1000) ? ((b.w >> 8) & 15) : k;      // qrec_bpr_exact_schedule_reg carries the slot in word 7
            if (to_slot < slots) {
                const int64_t to = ((int64_t)(s + lead) * slots + to_slot) * 2;
                wide[to] = a; wide[to + 1] = b;
            }
        }
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

template <typename T, int EPG>
int launch_levels_wide(void *P, void *Q, int64_t p_rows, int64_t q_rows, const int32_t *wide, int n_steps, int slots, double lr,
                       double regU, double regI, void *xlog, int64_t n, hipStream_t st) {
    const T tlr = (T)lr, cu = (T)lr * (T)regU, ci = (T)lr * (T)regI;
    const int nw = slots / 4;
#ifdef QREC_EXACT_PROBE      // build of tools/probe_exact_dbg.py only (make PROBE=1): timing variants with pieces removed -- their RESULTS ARE WRONG
    const char *dbg = getenv("QREC_EXACT_DBG");
    const int code = (dbg && sizeof(T) == 8 && EPG == 4) ? atoi(dbg) : 0;
#endif
    const size_t row_bytes = (size_t)16 * EPG * sizeof(T);
    QREC_REQUIRE((size_t)p_rows * row_bytes < 0xFFFFFF00ull && (size_t)q_rows * row_bytes < 0xFFFFFF00ull && (size_t)n * sizeof(T) < 0xFFFFFF00ull,
                 "qrec_bpr_sgd_scheduled_wide: the tables are addressed with 32-bit offsets (each below 4 GiB; use qrec_bpr_sgd_scheduled)");
    const uint32_t pb = (uint32_t)(p_rows * row_bytes), qb = (uint32_t)(q_rows * row_bytes), xb = (uint32_t)(n * sizeof(T));
#define QREC_REG_LAUNCH(C)                                                                                                     \
    hipLaunchKernelGGL((bpr_levels_reg_kernel<T, EPG, C>), dim3(1), dim3(64 * nw), 0, st, (T *)P, (T *)Q, pb, qb, (const int4 *)wide,   \
                       n_steps + QREC_EXACT_WIDE_LEAD, tlr, cu, ci, (T *)xlog, xb);
#ifdef QREC_EXACT_PROBE
    if constexpr (sizeof(T) == 8 && EPG == 4) {      // timing experiments (results are wrong): QREC_EXACT_DBG, see the kernel
        switch (code) {
            case 1: QREC_REG_LAUNCH(1) break;
            case 2: QREC_REG_LAUNCH(2) break;
            case 3: QREC_REG_LAUNCH(3) break;
            case 4: QREC_REG_LAUNCH(4) break;
            case 7: QREC_REG_LAUNCH(7) break;
            case 8: QREC_REG_LAUNCH(8) break;
            default: QREC_REG_LAUNCH(0)
        }
    } else {
        QREC_REG_LAUNCH(0)
    }
#else        // the product build has no such variants: a stale environment variable cannot corrupt an exact-mode run
    QREC_REG_LAUNCH(0)
#endif
#undef QREC_REG_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

// Which kernel executes an order-exact epoch: 1 = four triplets per wavefront (rows of 16 * {1, 2, 4, 8} elements: 16-byte
// accesses), 0 = one per wavefront (every wider row; QREC_EXACT_KERNEL=w64 forces it: the comparison instrument of
// tools/probe_exact.py)
int exact_kind(int ld) {
    const char *env = getenv("QREC_EXACT_KERNEL");
    if (env && !strcmp(env, "w64")) return 0;
    return (ld == 16 || ld == 32 || ld == 64 || ld == 128) ? 1 : 0;
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

int qrec_bpr_exact_kind(int dtype, int32_t ld, int32_t width, int32_t *kind, int32_t *slots) {
    QREC_REQUIRE(kind && slots && (dtype == QREC_F32 || dtype == QREC_F64) && width >= 1 && width <= QREC_EXACT_MAX_WIDTH,
                 "qrec_bpr_exact_kind: bad arguments");
    *kind = exact_kind(ld);
    *slots = *kind ? 4 * ((width + 3) / 4) : 0;
    return QREC_OK;
}

int qrec_bpr_exact_expand(const int32_t *d_entries, const int32_t *d_step_off, int64_t n_steps, int32_t slots, int32_t *d_wide,
                          void *stream) {
    QREC_REQUIRE(d_wide && n_steps >= 0 && n_steps < (1ll << 31) - 64 && slots >= 4 && slots <= QREC_EXACT_MAX_WIDTH && slots % 4 == 0,
                 "qrec_bpr_exact_expand: bad arguments");
    hipStream_t st = as_stream(stream);
    QREC_HIP_CHECK(hipMemsetAsync(d_wide, 0xFF, (size_t)(n_steps + QREC_EXACT_WIDE_PAD) * slots * 32, st));
    if (n_steps == 0) return QREC_OK;
    QREC_REQUIRE(d_entries && d_step_off, "qrec_bpr_exact_expand: null schedule");
    const int64_t blocks = std::min<int64_t>((n_steps + 15) / 16, 4096);
    // the layout starts with QREC_EXACT_WIDE_LEAD empty steps (the kernel has no prologue)
    hipLaunchKernelGGL(expand_schedule_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const int4 *)d_entries, d_step_off, (int)n_steps,
                       slots, QREC_EXACT_WIDE_LEAD, (int4 *)d_wide);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_bpr_sgd_scheduled_wide(void *d_P, void *d_Q, int64_t n_users, int64_t n_items, int dtype, int32_t d, int32_t ld,
                                const int32_t *d_wide, int64_t n_steps, int32_t slots, int64_t n, double lr, double regU, double regI,
                                void *d_xlog, double *d_scratch, double *d_loss, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_loss && n >= 0 && n_steps >= 0 && n_steps < (1ll << 31) - 64 && n_users >= 0 && n_items >= 0,
                 "qrec_bpr_sgd_scheduled_wide: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_bpr_sgd_scheduled_wide: bad dtype %d", dtype);
    QREC_REQUIRE(d >= 1 && ld >= d && (ld == 16 || ld == 32 || ld == 64 || ld == 128),
                 "qrec_bpr_sgd_scheduled_wide: rows of 16, 32, 64 or 128 elements (got d=%d ld=%d)", d, ld);
    QREC_REQUIRE(slots >= 4 && slots % 4 == 0 && slots <= QREC_EXACT_MAX_WIDTH, "qrec_bpr_sgd_scheduled_wide: slots must be 4, 8, 12 or 16");
    hipStream_t st = as_stream(stream);
    if (n == 0) { QREC_HIP_CHECK(hipMemsetAsync(d_loss, 0, sizeof(double), st)); return QREC_OK; }
    QREC_REQUIRE(d_wide && d_xlog && d_scratch, "qrec_bpr_sgd_scheduled_wide: null schedule / scratch");
    int rc;
#define QREC_WIDE(T) (ld == 16 ? launch_levels_wide<T, 1> : ld == 32 ? launch_levels_wide<T, 2> : ld == 64 ? launch_levels_wide<T, 4> : launch_levels_wide<T, 8>)
    rc = dtype == QREC_F64 ? QREC_WIDE(double)(d_P, d_Q, n_users, n_items, d_wide, (int)n_steps, slots, lr, regU, regI, d_xlog, n, st)
                           : QREC_WIDE(float)(d_P, d_Q, n_users, n_items, d_wide, (int)n_steps, slots, lr, regU, regI, d_xlog, n, st);
#undef QREC_WIDE
    if (rc != QREC_OK) return rc;
    unsigned int *ticket = reinterpret_cast<unsigned int *>(d_scratch + kNllBlocks);
    if (dtype == QREC_F64)
        hipLaunchKernelGGL((nll_sum_kernel<double>), dim3(kNllBlocks), dim3(256), 0, st, (const double *)d_xlog, n, d_scratch, ticket, d_loss);
    else
        hipLaunchKernelGGL((nll_sum_kernel<float>), dim3(kNllBlocks), dim3(256), 0, st, (const float *)d_xlog, n, d_scratch, ticket, d_loss);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
