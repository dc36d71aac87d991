// BPR SGD on embedding rows: model/ranking/BPR.py:45-53 (+ sigmoid util/qmath.py:127-128).
//
//   x = P[u].Q[i] - P[u].Q[j];  s = 1/(1+exp(-x));  g = lr*(1-s)
//   P[u] += g*(Q[i]-Q[j]);  Q[i] += g*P[u];  Q[j] -= g*P[u]      (P[u] already updated)
//   P[u] -= lr*regU*P[u];   Q[i] -= lr*regI*Q[i];   Q[j] -= lr*regI*Q[j];   loss += -log(s)
//
// Two kernels:
//  * bpr_ordered_kernel  -- the reference's semantics: triplets applied strictly in array
//    order.  The conflict DAG of an epoch is ~n/6 deep (every user run is a chain through
//    P[u], popular items link the runs), so the order-exact mode has no exploitable
//    parallelism beyond the d lanes of one row; ONE wavefront walks the list, with the
//    next triplet's rows prefetched and patched from registers when they alias.  This is
//    the parity mode (fp64 or fp32).
//  * bpr_hogwild_kernel  -- throughput mode.  HBM/L2-latency bound gather + scatter:
//    a group of LPR lanes (16 for d<=64: lane r holds columns r, r+16, r+32, r+48 of one
//    256-B row, 4 rows per wavefront) owns a chunk of consecutive triplets, keeps P[u] in
//    registers along the user run, and applies every row update as the exact per-sample
//    delta (atomic f32 add, so concurrent chunks never lose an update).  Index tiles are
//    staged through LDS per wavefront; dot products reduce inside the LPR-lane row.
//    Algorithmic bytes per triplet: 6*d*4 + 12.
#include <cmath>

#include <algorithm>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

using namespace qrec;

namespace {

// ------------------------------------------------------------------------------------
// order-exact kernel
// ------------------------------------------------------------------------------------
template <typename T> __device__ inline T dev_exp(T x);
template <> __device__ inline float dev_exp<float>(float x) { return expf(x); }
template <> __device__ inline double dev_exp<double>(double x) { return exp(x); }

// -log(sigmoid(x)) = softplus(-x), evaluated without forming sigmoid first: in fp32 sigmoid(x)
// underflows to 0 for x < -88.7 and -log(0) = inf, while the reference's fp64 value (= -x) is
// finite down to x ~ -745.  Same value to rounding wherever the direct form is finite.
__device__ inline float neg_log_sigmoid(float x) {
    return x >= 0.f ? log1pf(expf(-x)) : (-x + log1pf(expf(x)));
}
__device__ inline double neg_log_sigmoid(double x) {
    return x >= 0.0 ? log1p(exp(-x)) : (-x + log1p(exp(x)));
}

template <typename T>
__device__ inline T wave_allreduce_sum(T v) { return wave_sum_dpp(v); }   // common.h: DPP, not the LDS crossbar

// EPL = elements per lane; lane l owns columns l, l+64, ...  (d <= 64*EPL)
template <typename T, int EPL>
__global__ __launch_bounds__(64) void bpr_ordered_kernel(
    T *__restrict__ P, T *__restrict__ Q, int d, int ld, const int32_t *__restrict__ u_idx,
    const int32_t *__restrict__ i_idx, const int32_t *__restrict__ j_idx, int64_t n, T lr, T cu,
    T ci, double *__restrict__ loss_out) {
#pragma clang fp contract(off)  // numpy rounds every product and sum separately
    const int lane = threadIdx.x;
    bool valid[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) valid[e] = (lane + 64 * e) < d;

    auto load_row = [&](const T *tab, int row, T (&dst)[EPL]) {
        const T *p = tab + (int64_t)row * ld + lane;
#pragma unroll
        for (int e = 0; e < EPL; e++) dst[e] = valid[e] ? p[64 * e] : T(0);
    };
    auto store_row = [&](T *tab, int row, const T (&src)[EPL]) {
        T *p = tab + (int64_t)row * ld + lane;
#pragma unroll
        for (int e = 0; e < EPL; e++)
            if (valid[e]) p[64 * e] = src[e];
    };

    T pu[EPL], qi[EPL], qj[EPL], nqi[EPL], nqj[EPL];
    double loss = 0.0;
    int cur_u = -1;
    if (n > 0) { load_row(Q, i_idx[0], qi); load_row(Q, j_idx[0], qj); }
    for (int64_t t = 0; t < n; t++) {
        const int ut = u_idx[t], it = i_idx[t], jt = j_idx[t];
        if (ut != cur_u) {
            if (cur_u >= 0) store_row(P, cur_u, pu);
            load_row(P, ut, pu);
            cur_u = ut;
        }
        int in = -1, jn = -1;
        if (t + 1 < n) {  // prefetch the next triplet's item rows under this one's math
            in = i_idx[t + 1]; jn = j_idx[t + 1];
            load_row(Q, in, nqi); load_row(Q, jn, nqj);
        }
        T di = 0, dj = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) { di += pu[e] * qi[e]; dj += pu[e] * qj[e]; }
        di = wave_allreduce_sum(di); dj = wave_allreduce_sum(dj);
        const T s = T(1) / (T(1) + dev_exp<T>(-(di - dj)));
        const T g = lr * (T(1) - s);
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            pu[e] += g * (qi[e] - qj[e]);
            qi[e] += g * pu[e];
            qj[e] -= g * pu[e];
            pu[e] -= cu * pu[e];
            qi[e] -= ci * qi[e];
            qj[e] -= ci * qj[e];
        }
        store_row(Q, it, qi); store_row(Q, jt, qj);
        // fp64 tables: the reference's own expression (model/ranking/BPR.py:53); fp32: stable form
        if constexpr (sizeof(T) == 8) loss += -log((double)s);
        else loss += neg_log_sigmoid((double)(di - dj));
        if (t + 1 < n) {  // rows just written supersede what the prefetch saw
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                T a = nqi[e], b = nqj[e];
                if (in == it) a = qi[e]; else if (in == jt) a = qj[e];
                if (jn == it) b = qi[e]; else if (jn == jt) b = qj[e];
                qi[e] = a; qj[e] = b;
            }
        }
    }
    if (cur_u >= 0) store_row(P, cur_u, pu);
    if (lane == 0) *loss_out = loss;
}

// TBPR (model/ranking/TBPR.py:40-48,131-159), order-exact: the same per-triplet arithmetic over the chained
// (u, a, b) updates, with two things BPR never needs:
//   * a == b happens (the closing random item may repeat the chain's last social item): the two row updates then hit
//     ONE row in sequence, as numpy's in-place row statements do;
//   * the loss takes regU*sum(P*P) + regI*sum(Q*Q) over the WHOLE tables after every user (TBPR.py:159 is inside the
//     user loop).  The sums are carried along instead of recomputed: sums_in = {sum P*P, sum Q*Q} before the launch,
//     every row update adds (|new row|^2 - |old row|^2) in fp64.
// loss_out[0] = sum(-log s), loss_out[1] = sum over users of (regU*sumP2 + regI*sumQ2).
template <typename T, int EPL>
__global__ __launch_bounds__(64) void tbpr_ordered_kernel(
    T *__restrict__ P, T *__restrict__ Q, int d, int ld, const int32_t *__restrict__ u_idx,
    const int32_t *__restrict__ a_idx, const int32_t *__restrict__ b_idx, int64_t n, T lr, T cu, T ci,
    double regU, double regI, const double *__restrict__ sums_in, double *__restrict__ loss_out) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    bool valid[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) valid[e] = (lane + 64 * e) < d;
    auto load_row = [&](const T *tab, int row, T (&dst)[EPL]) {
        const T *p = tab + (int64_t)row * ld + lane;
#pragma unroll
        for (int e = 0; e < EPL; e++) dst[e] = valid[e] ? p[64 * e] : T(0);
    };
    auto store_row = [&](T *tab, int row, const T (&src)[EPL]) {
        T *p = tab + (int64_t)row * ld + lane;
#pragma unroll
        for (int e = 0; e < EPL; e++)
            if (valid[e]) p[64 * e] = src[e];
    };
    auto norm2 = [&](const T (&r)[EPL]) {
        double v = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; e++) v += (double)r[e] * (double)r[e];
        return wave_allreduce_sum(v);
    };
    T pu[EPL], qa[EPL], qb[EPL], nqa[EPL], nqb[EPL];
    double loss = 0.0, reg = 0.0, sP = sums_in[0], sQ = sums_in[1], pu_loaded = 0.0;
    int cur_u = -1;
    if (n > 0) { load_row(Q, a_idx[0], qa); load_row(Q, b_idx[0], qb); }
    for (int64_t t = 0; t < n; t++) {
        const int ut = u_idx[t], at = a_idx[t], bt = b_idx[t];
        if (ut != cur_u) {
            if (cur_u >= 0) {
                store_row(P, cur_u, pu);
                sP += norm2(pu) - pu_loaded;
                reg += regU * sP + regI * sQ;
            }
            load_row(P, ut, pu);
            pu_loaded = norm2(pu);
            cur_u = ut;
        }
        int an = -1, bn = -1;
        if (t + 1 < n) { an = a_idx[t + 1]; bn = b_idx[t + 1]; load_row(Q, an, nqa); load_row(Q, bn, nqb); }
        const bool alias = at == bt;
        const double q_old = norm2(qa) + (alias ? 0.0 : norm2(qb));
        T da = 0, db = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) { da += pu[e] * qa[e]; db += pu[e] * qb[e]; }
        da = wave_allreduce_sum(da); db = wave_allreduce_sum(db);
        const T s = T(1) / (T(1) + dev_exp<T>(-(da - db)));
        const T g = lr * (T(1) - s);
        if (!alias) {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                pu[e] += g * (qa[e] - qb[e]);
                qa[e] += g * pu[e];
                qb[e] -= g * pu[e];
                pu[e] -= cu * pu[e];
                qa[e] -= ci * qa[e];
                qb[e] -= ci * qb[e];
            }
            store_row(Q, at, qa); store_row(Q, bt, qb);
        } else {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                T r = qa[e];
                pu[e] += g * (r - r);
                r += g * pu[e];
                r -= g * pu[e];
                pu[e] -= cu * pu[e];
                r -= ci * r;
                r -= ci * r;
                qa[e] = r; qb[e] = r;
            }
            store_row(Q, at, qa);
        }
        sQ += norm2(qa) + (alias ? 0.0 : norm2(qb)) - q_old;
        if constexpr (sizeof(T) == 8) loss += -log((double)s);
        else loss += neg_log_sigmoid((double)(da - db));
        if (t + 1 < n) {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                T x = nqa[e], y = nqb[e];
                if (an == at) x = qa[e]; else if (an == bt) x = qb[e];
                if (bn == at) y = qa[e]; else if (bn == bt) y = qb[e];
                qa[e] = x; qb[e] = y;
            }
        }
    }
    if (cur_u >= 0) {
        store_row(P, cur_u, pu);
        sP += norm2(pu) - pu_loaded;
        reg += regU * sP + regI * sQ;
    }
    if (lane == 0) { loss_out[0] = loss; loss_out[1] = reg; }
}

// SBPR (model/ranking/SBPR.py:41-74, numpy path), order-exact, one wavefront, lane = column.  Rows (u, i, k, j, Suk) in the
// reference's visiting order; k < 0: the user has no social feedback (:68-73, plain BPR step with the item biases in the
// score and NO decay); i < 0: a bare visit of a user without positives (only the per-user loss terms).  With k >= 0 the two
// chained pairs of :45-55 -- (i over k) scaled by 1 / (Suk + 1), then (k over j) -- then the four decays (:56-59, in the
// order P[u], Q[i], Q[j], Q[k]) and the loss of :57-58 from the UPDATED rows, without the biases.  j == k happens (the
// negative's rejection test does not look at the user's own feedback list): the row is then updated in sequence, as numpy's
// in-place statements do, and decays twice.  The biases b never change (no statement updates them).  After every user:
// regU*sum(P*P) + regI*sum(Q*Q) + b.b over the WHOLE tables (:74 sits inside the user loop) -- sums carried along as in
// tbpr_ordered_kernel.  loss_out[0] = sum of the -log terms, loss_out[1] = sum over users of the table terms.
template <typename T, int EPL>
__global__ __launch_bounds__(64) void sbpr_ordered_kernel(
    T *__restrict__ P, T *__restrict__ Q, const T *__restrict__ bias, int d, int ld, const int32_t *__restrict__ rows, int64_t n, T lr, T cu, T ci,
    double regU, double regI, double bb, const double *__restrict__ sums_in, double *__restrict__ loss_out) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    bool valid[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) valid[e] = (lane + 64 * e) < d;
    auto load_row = [&](const T *tab, int row, T (&dst)[EPL]) {
        const T *p = tab + (int64_t)row * ld + lane;
#pragma unroll
        for (int e = 0; e < EPL; e++) dst[e] = valid[e] ? p[64 * e] : T(0);
    };
    auto store_row = [&](T *tab, int row, const T (&src)[EPL]) {
        T *p = tab + (int64_t)row * ld + lane;
#pragma unroll
        for (int e = 0; e < EPL; e++)
            if (valid[e]) p[64 * e] = src[e];
    };
    auto norm2 = [&](const T (&r)[EPL]) {
        double v = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; e++) v += (double)r[e] * (double)r[e];
        return wave_allreduce_sum(v);
    };
    auto dot = [&](const T (&a)[EPL], const T (&b)[EPL]) {
        T v = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) v += a[e] * b[e];
        return wave_allreduce_sum(v);
    };
    // a pairwise step on (pu, hi, lo) with coefficient c: P[u] += c (hi - lo); hi += c P[u]; lo -= c P[u]   (:46-48, :53-55, :70-72)
    auto pair_step = [&](T (&pu)[EPL], T (&hi)[EPL], T (&lo)[EPL], T c) {
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            pu[e] += c * (hi[e] - lo[e]);
            hi[e] += c * pu[e];
            lo[e] -= c * pu[e];
        }
    };
    auto decay = [&](T (&r)[EPL], T c) {
#pragma unroll
        for (int e = 0; e < EPL; e++) r[e] -= c * r[e];
    };
    auto sigmoid = [&](T x) { return T(1) / (T(1) + dev_exp<T>(-x)); };
    T pu[EPL], qi[EPL], qk[EPL], qj[EPL];
    double loss = 0.0, reg = 0.0, sP = sums_in[0], sQ = sums_in[1], pu_loaded = 0.0;
    int cur_u = -1;
    for (int64_t t = 0; t < n; t++) {
        const int32_t *row = rows + 5 * t;
        const int ut = row[0], it = row[1], kt = row[2], jt = row[3];
        const T scale = T(1) / (T)(row[4] + 1);
        if (ut != cur_u) {
            if (cur_u >= 0) {
                store_row(P, cur_u, pu);
                sP += norm2(pu) - pu_loaded;
                reg += regU * sP + regI * sQ + bb;
            }
            load_row(P, ut, pu);
            pu_loaded = norm2(pu);
            cur_u = ut;
        }
        if (it < 0) continue;
        load_row(Q, it, qi); load_row(Q, jt, qj);
        if (kt < 0) {
            const double q_old = norm2(qi) + norm2(qj);
            const T x = ((dot(pu, qi) - dot(pu, qj)) + bias[it]) - bias[jt];
            const T s = sigmoid(x);
            pair_step(pu, qi, qj, lr * (T(1) - s));
            store_row(Q, it, qi); store_row(Q, jt, qj);
            sQ += norm2(qi) + norm2(qj) - q_old;
            if constexpr (sizeof(T) == 8) loss += -log((double)s);
            else loss += neg_log_sigmoid((double)x);      // fp32 tables: the stable form of -log(sigmoid(x)) (bpr_ordered_kernel)
            continue;
        }
        const bool alias = jt == kt;
        load_row(Q, kt, qk);
        const double q_old = norm2(qi) + norm2(qk) + (alias ? 0.0 : norm2(qj));
        const T s = sigmoid((((dot(pu, qi) - dot(pu, qk)) + bias[it]) - bias[kt]) / (T)(row[4] + 1));     // the reference divides (:45) ...
        pair_step(pu, qi, qk, (scale * lr) * (T(1) - s));                                  // ... and multiplies here: 1 / (Suk+1) * lRate * (1 - s), left to right (:46)
        if (alias) {
            const T s2 = sigmoid(((dot(pu, qk) - dot(pu, qk)) + bias[kt]) - bias[jt]);
            const T c2 = lr * (T(1) - s2);
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                pu[e] += c2 * (qk[e] - qk[e]);
                qk[e] += c2 * pu[e];
                qk[e] -= c2 * pu[e];
            }
            decay(pu, cu); decay(qi, ci); decay(qk, ci); decay(qk, ci);
        } else {
            const T s2 = sigmoid(((dot(pu, qk) - dot(pu, qj)) + bias[kt]) - bias[jt]);
            pair_step(pu, qk, qj, lr * (T(1) - s2));
            decay(pu, cu); decay(qi, ci); decay(qj, ci); decay(qk, ci);
        }
        const T x1 = (dot(pu, qi) - dot(pu, qk)) / (T)(row[4] + 1);
        const T x2 = alias ? (dot(pu, qk) - dot(pu, qk)) : (dot(pu, qk) - dot(pu, qj));
        if constexpr (sizeof(T) == 8) loss += -log((double)sigmoid(x1)) - log((double)sigmoid(x2));
        else loss += neg_log_sigmoid((double)x1) + neg_log_sigmoid((double)x2);
        store_row(Q, it, qi); store_row(Q, kt, qk);
        if (!alias) store_row(Q, jt, qj);
        sQ += norm2(qi) + norm2(qk) + (alias ? 0.0 : norm2(qj)) - q_old;
    }
    if (cur_u >= 0) {
        store_row(P, cur_u, pu);
        sP += norm2(pu) - pu_loaded;
        reg += regU * sP + regI * sQ + bb;
    }
    if (lane == 0) { loss_out[0] = loss; loss_out[1] = reg; }
}

// Rating-prediction MF family, order-exact, same structure (one wavefront, lane = column):
//   VAR 0  model/rating/BasicMF.py:9-26   P[u] += (lr*e)*q ;            Q[i] += (lr*e)*p
//   VAR 1  model/rating/PMF.py:9-28       P[u] += lr*(e*q - regU*p) ;   Q[i] += lr*(e*p - regI*q)
//   VAR 2  model/rating/SVD.py:13-35      as PMF with e = r - (((P[u].Q[i] + mean) + Bi[i]) + Bu[u]),
//                                         Bu[u] += lr*(e - regB*bu) ; Bi[i] += lr*(e - regB*bi)
//   VAR 3  model/rating/EE.py:15-34       diff = P[u]-Q[i]; e = r - (((mean + Bi[i]) + Bu[u]) - diff.diff);
//                                         loss += e*e + regU*diff.diff; P[u] -= (lr*(e+regU))*diff;
//                                         Q[i] += (lr*(e+regI))*(P[u]-Q[i]); biases as SVD
// p is a VIEW of P[u] in the reference, so the Q[i] update sees the already updated P[u].
template <typename T, int EPL, int VAR>
__global__ __launch_bounds__(64) void mf_ordered_kernel(
    T *__restrict__ P, T *__restrict__ Q, T *__restrict__ Bu, T *__restrict__ Bi, int d, int ld,
    const int32_t *__restrict__ u_idx, const int32_t *__restrict__ i_idx, const double *__restrict__ rating,
    int64_t n, T lr, T regU, T regI, T regB, T gmean, double *__restrict__ loss_out) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    double loss = 0.0;
    for (int64_t t = 0; t < n; t++) {
        const int u = u_idx[t], i = i_idx[t];
        T *p = P + (int64_t)u * ld + lane;
        T *q = Q + (int64_t)i * ld + lane;
        T pv[EPL], qv[EPL], dot = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const bool ok = (lane + 64 * e) < d;
            pv[e] = ok ? p[64 * e] : T(0);
            qv[e] = ok ? q[64 * e] : T(0);
            if constexpr (VAR == 3) { const T df = pv[e] - qv[e]; dot += df * df; }
            else dot += pv[e] * qv[e];
        }
        dot = wave_allreduce_sum(dot);
        T pred = dot, bu = 0, bi = 0;
        if constexpr (VAR == 2) { bu = Bu[u]; bi = Bi[i]; pred = ((dot + gmean) + bi) + bu; }
        if constexpr (VAR == 3) { bu = Bu[u]; bi = Bi[i]; pred = ((gmean + bi) + bu) - dot; }
        const T err = (T)rating[t] - pred;
        loss += (double)err * (double)err;
        if constexpr (VAR == 3) loss += (double)(regU * dot);
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            if constexpr (VAR == 0) {
                pv[e] += (lr * err) * qv[e];
                qv[e] += (lr * err) * pv[e];
            } else if constexpr (VAR == 3) {
                pv[e] -= (lr * (err + regU)) * (pv[e] - qv[e]);
                qv[e] += (lr * (err + regI)) * (pv[e] - qv[e]);
            } else {
                pv[e] += lr * (err * qv[e] - regU * pv[e]);
                qv[e] += lr * (err * pv[e] - regI * qv[e]);
            }
            if ((lane + 64 * e) < d) { p[64 * e] = pv[e]; q[64 * e] = qv[e]; }
        }
        if constexpr (VAR >= 2) {
            if (lane == 0) { Bu[u] = bu + lr * (err - regB * bu); Bi[i] = bi + lr * (err - regB * bi); }
        }
    }
    if (lane == 0) *loss_out = loss;
}

// model/rating/SVDPlusPlus.py:25-62,70-86, order-exact (one wavefront, lane = column).  Per rating the user's
// rated items (data.userRated order, CSR) are walked twice: sum_j Y[j] for the prediction, then -- over the OTHER
// items -- their sum, the Y updates and the implicit-feedback term of Q[i].  p and q are views in the reference: the
// P[u] update sees the already updated Q[i], the final Q[i] update the updated P[u].
template <typename T, int EPL>
__global__ __launch_bounds__(64) void svdpp_ordered_kernel(
    T *__restrict__ P, T *__restrict__ Q, T *__restrict__ Y, T *__restrict__ Bu, T *__restrict__ Bi, int d, int ld,
    const int64_t *__restrict__ rated_indptr, const int32_t *__restrict__ rated_items,
    const int32_t *__restrict__ u_idx, const int32_t *__restrict__ i_idx, const double *__restrict__ rating, int64_t n,
    T lr, T regU, T regI, T regB, T regY, T gmean, double *__restrict__ loss_out) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    double loss = 0.0;
    for (int64_t t = 0; t < n; t++) {
        const int u = u_idx[t], i = i_idx[t];
        T *p = P + (int64_t)u * ld + lane;
        T *q = Q + (int64_t)i * ld + lane;
        const int64_t b = rated_indptr[u], e_ = rated_indptr[u + 1];
        const T w = (T)(e_ - b);
        T pv[EPL], qv[EPL], sum[EPL];
        bool ok[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            ok[e] = (lane + 64 * e) < d;
            pv[e] = ok[e] ? p[64 * e] : T(0);
            qv[e] = ok[e] ? q[64 * e] : T(0);
            sum[e] = T(0);
        }
        for (int64_t k = b; k < e_; k++) {
            const T *y = Y + (int64_t)rated_items[k] * ld + lane;
#pragma unroll
            for (int e = 0; e < EPL; e++) sum[e] += ok[e] ? y[64 * e] : T(0);
        }
        T a = 0, dot = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            if (e_ > b) a += (sum[e] / w) * qv[e];
            dot += pv[e] * qv[e];
        }
        a = wave_allreduce_sum(a);
        dot = wave_allreduce_sum(dot);
        const T bu = Bu[u], bi = Bi[i];
        const T pred = a + (((dot + gmean) + bi) + bu);
        const T err = (T)rating[t] - pred;
        loss += (double)err * (double)err;
        if (lane == 0) { Bu[u] = bu + lr * (err - regB * bu); Bi[i] = bi + lr * (err - regB * bi); }
        if (e_ - b > 1) {
            const T wm1 = (T)(e_ - b - 1);
            bool first = true;
            for (int64_t k = b; k < e_; k++) {
                const int j = rated_items[k];
                if (j == i) continue;
                T *y = Y + (int64_t)j * ld + lane;
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const T yv = ok[e] ? y[64 * e] : T(0);
                    sum[e] = first ? yv : sum[e] + yv;
                    if (ok[e]) y[64 * e] = yv + lr * ((err * qv[e]) / wm1 - regY * yv);
                }
                first = false;
            }
            if (!first) {
#pragma unroll
                for (int e = 0; e < EPL; e++) qv[e] += ((lr * err) * sum[e]) / wm1;
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            pv[e] += lr * (err * qv[e] - regU * pv[e]);
            qv[e] += lr * (err * pv[e] - regI * qv[e]);
            if (ok[e]) { p[64 * e] = pv[e]; q[64 * e] = qv[e]; }
        }
    }
    if (lane == 0) *loss_out = loss;
}

// ------------------------------------------------------------------------------------
// throughput kernel
// ------------------------------------------------------------------------------------
// Register layout: a group of LPR lanes owns one row of ld = LPR*E floats; lane r holds
// columns r, r+LPR, ..., r+(E-1)*LPR.  Every load / atomic instruction therefore touches
// one CONTIGUOUS 4*LPR-byte segment per group (64 B for d=64).  Measured on MI355X
// (tools/ubench/atomics.hip): the L2 atomic units retire ~1 dword/clk/channel when a
// request covers a contiguous 64-B segment, 4x less when the same dwords are strided by
// 16 B (the float4-per-lane layout) -- atomic cost is per request, not per byte.
constexpr int kMaxChunk = 64;
constexpr int64_t kBufLimit = (int64_t)1 << 32;   // a buffer descriptor addresses 32-bit byte offsets
// test hook (README switch table): QREC_FORCE_64BIT_ADDRESSING=1 sends tables of any size down the >= 4 GiB flavour (TabPtr), so that
// its parity tests and timings do not need a 4 GiB table
static inline bool force_64bit_addressing() { const char *e = getenv("QREC_FORCE_64BIT_ADDRESSING"); return e && e[0] == '1'; }

enum : int { LD_PLAIN = 0, LD_SC1 = 1 };
enum : int { UP_STORE = 0, UP_STORE_SC1 = 1, UP_ATOMIC = 2 };

template <int E> struct Row { float v[E]; };

// Row access of the throughput kernels, two flavours chosen by table size:
//   TabBuf  a buffer descriptor sized to the table (32-bit byte offsets, so < 4 GiB): out-of-range row ids are
//           dropped by the hardware bounds check instead of corrupting memory, and address arithmetic is one VALU op;
//   TabPtr  64-bit global addressing for tables of 4 GiB and more (measured 2-5% slower: tools/bench_configs.py).
// kPrefetchAlways: how the triplet loops issue the next triplet's row loads (see the item-major kernel)
struct TabBuf {
    static constexpr bool kPrefetchAlways = false;
    __amdgpu_buffer_rsrc_t rs;
    static __device__ inline TabBuf make(float *p, int64_t bytes) { return TabBuf{make_rsrc(p, (uint32_t)bytes)}; }
};
struct TabPtr {
    static constexpr bool kPrefetchAlways = true;
    float *base;
    static __device__ inline TabPtr make(float *p, int64_t) { return TabPtr{p}; }
};

template <int LPR, int E, int LOADP>
__device__ inline Row<E> hw_load_row(TabBuf t, int row, int r) {
    constexpr int aux = (LOADP == LD_SC1) ? kAuxSc1 : kAuxPlain;
    const uint32_t off = ((uint32_t)row * (uint32_t)(LPR * E) + (uint32_t)r) * 4u;
    Row<E> out;
#pragma unroll
    for (int e = 0; e < E; e++)
        out.v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rs, (int)(off + 4u * LPR * e), 0, aux));
    return out;
}
template <int LPR, int E, int LOADP>
__device__ inline Row<E> hw_load_row(TabPtr t, int row, int r) {
    const float *p = t.base + (int64_t)row * (LPR * E) + r;
    Row<E> out;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if constexpr (LOADP == LD_SC1) out.v[e] = __hip_atomic_load(p + LPR * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load ... sc1
        else out.v[e] = p[LPR * e];
    }
    return out;
}

template <int LPR, int E, int UPD>
__device__ inline void hw_update_row(TabBuf t, int row, int r, const Row<E> &oldv, const Row<E> &newv) {
    const uint32_t off = ((uint32_t)row * (uint32_t)(LPR * E) + (uint32_t)r) * 4u;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if constexpr (UPD == UP_ATOMIC) {
            // exact per-sample delta (new-old is exact when |delta| << |value|)
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(newv.v[e] - oldv.v[e], t.rs, (int)(off + 4u * LPR * e), 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, newv.v[e]), t.rs, (int)(off + 4u * LPR * e), 0,
                                                  UPD == UP_STORE_SC1 ? kAuxSc1 : kAuxPlain);
        }
    }
}
template <int LPR, int E, int UPD>
__device__ inline void hw_update_row(TabPtr t, int row, int r, const Row<E> &oldv, const Row<E> &newv) {
    float *p = t.base + (int64_t)row * (LPR * E) + r;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if constexpr (UPD == UP_ATOMIC) (void)__hip_atomic_fetch_add(p + LPR * e, newv.v[e] - oldv.v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if constexpr (UPD == UP_STORE_SC1) __hip_atomic_store(p + LPR * e, newv.v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[LPR * e] = newv.v[e];
    }
}

// Learning rate of a throughput epoch: either the host's value, or -- when `state` is given -- the device-resident
// bold-driver state written by epoch_close_kernel (reduce.hip; layout in include/qrec_hip.h), so that a whole
// training run can be enqueued without a host round trip per epoch.  A converged / failed run turns the
// remaining enqueued epochs into no-ops.
struct HwRate {
    float lr, regU, regI;
    const double *state;
};
__device__ inline bool hw_rate_resolve(const HwRate &rt, float &lr, float &cu, float &ci) {
    lr = rt.lr;
    if (rt.state) {
        if (rt.state[QREC_DRV_CONVERGED] != 0.0 || rt.state[QREC_DRV_FAILED] != 0.0) return false;
        lr = (float)rt.state[QREC_DRV_LR];
    }
    cu = lr * rt.regU; ci = lr * rt.regI;
    return true;
}

template <int LPR, int E, int LOADP, int UPD, typename TAB>
__global__ __launch_bounds__(256) void bpr_hogwild_kernel(
    float *__restrict__ P, float *__restrict__ Q, int64_t p_bytes, int64_t q_bytes,
    const int32_t *__restrict__ u_idx, const int32_t *__restrict__ i_idx,
    const int32_t *__restrict__ j_idx, int64_t n, int chunk, int64_t n_chunks,
    int64_t groups_active, HwRate rate, double *__restrict__ loss_out) {
    constexpr int GPW = kWave / LPR;  // groups (rows) per wavefront
    float lr, cu, ci;
    if (!hw_rate_resolve(rate, lr, cu, ci)) return;
    __shared__ int32_t s_idx[4][GPW][3][kMaxChunk];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + wave) * GPW + g;
    int32_t *su = s_idx[wave][g][0], *si = s_idx[wave][g][1], *sj = s_idx[wave][g][2];

    const TAB rsP = TAB::make(P, p_bytes), rsQ = TAB::make(Q, q_bytes);
    float loss = 0.f;
    double loss_acc = 0.0;

    for (int64_t c = gid; c < n_chunks && gid < groups_active; c += groups_active) {
        const int64_t t0 = c * chunk;
        const int len = (int)((n - t0) < chunk ? (n - t0) : chunk);
        // stage this group's (u,i,j) tile in LDS: coalesced 4*LPR-byte segments
        for (int k = r; k < len; k += LPR) {
            su[k] = u_idx[t0 + k]; si[k] = i_idx[t0 + k]; sj[k] = j_idx[t0 + k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        int cur_u = -1, it = si[0], jt = sj[0];
        Row<E> pu, pu0;
#pragma unroll
        for (int e = 0; e < E; e++) pu.v[e] = pu0.v[e] = 0.f;
        Row<E> qi = hw_load_row<LPR, E, LOADP>(rsQ, it, r);
        Row<E> qj = hw_load_row<LPR, E, LOADP>(rsQ, jt, r);
        for (int k = 0; k < len; k++) {
            const int ut = su[k];
            if (ut != cur_u) {
                if (cur_u >= 0) hw_update_row<LPR, E, UPD>(rsP, cur_u, r, pu0, pu);
                pu = hw_load_row<LPR, E, LOADP>(rsP, ut, r);
                pu0 = pu; cur_u = ut;
            }
            // next triplet's rows are in flight under this one's math (the form of the prefetch per addressing flavour: see the item-major kernel)
            const bool more = (k + 1 < len);
            int in = it, jn = jt;
            Row<E> nqi = qi, nqj = qj;
            if (TAB::kPrefetchAlways || more) {
                const int kn = more ? k + 1 : k;
                in = si[kn]; jn = sj[kn];
                nqi = hw_load_row<LPR, E, LOADP>(rsQ, in, r);
                nqj = hw_load_row<LPR, E, LOADP>(rsQ, jn, r);
            }
            float di = 0.f, dj = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) { di += pu.v[e] * qi.v[e]; dj += pu.v[e] * qj.v[e]; }
            di = row_allreduce_sum<LPR>(di); dj = row_allreduce_sum<LPR>(dj);
            const float s = 1.0f / (1.0f + expf(-(di - dj)));
            const float gsc = lr * (1.0f - s);
            Row<E> qin, qjn;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float pun = pu.v[e] + gsc * (qi.v[e] - qj.v[e]);
                float a = qi.v[e] + gsc * pun, b = qj.v[e] - gsc * pun;
                qin.v[e] = a - ci * a; qjn.v[e] = b - ci * b;
                pu.v[e] = pun - cu * pun;
            }
            hw_update_row<LPR, E, UPD>(rsQ, it, r, qi, qin);
            hw_update_row<LPR, E, UPD>(rsQ, jt, r, qj, qjn);
            loss += neg_log_sigmoid(di - dj);
            if (more) {  // rows this group just changed supersede the prefetched copy
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (in == it) nqi.v[e] = qin.v[e]; else if (in == jt) nqi.v[e] = qjn.v[e];
                    if (jn == it) nqj.v[e] = qin.v[e]; else if (jn == jt) nqj.v[e] = qjn.v[e];
                }
            }
            qi = nqi; qj = nqj; it = in; jt = jn;
        }
        if (cur_u >= 0) hw_update_row<LPR, E, UPD>(rsP, cur_u, r, pu0, pu);
        loss_acc += (double)loss; loss = 0.f;  // <=64 fp32 terms per chunk, fp64 across chunks
        __builtin_amdgcn_wave_barrier();       // tile is re-staged by the next chunk
    }
    if (r != 0) loss_acc = 0.0;  // every lane of a group carries the same value
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) loss_acc += __shfl_xor(loss_acc, m, kWave);
    if (lane == 0 && loss_acc != 0.0) atomicAdd(loss_out, loss_acc);  // one f64 atomic per wavefront
}

// ------------------------------------------------------------------------------------
// Item-major schedule of the same Hogwild epoch.
// The triplets arrive sorted by positive item; a group keeps Q[i] in registers along an item run
// (exact chain on that row), and applies the per-sample deltas of P[u] and Q[j] atomically.
// Why: the atomic units serialise same-line requests.  In the user-major schedule the per-triplet
// atomics land on Q[i] ~ Zipf(0.6) (6.3k touches on the top item per epoch) and Q[j] ~ uniform:
// 3.8 G row-updates/s.  Item-major moves them to P[u] ~ Zipf(0.4) (max 1.35k) and Q[j]: 4.8 G/s,
// the uniform-address rate (tools/ubench/atomics3.hip).  Q[i] itself is flushed as one delta every
// `flush_every` triplets and re-read, so that concurrent runs of the same hot item see each other.
// Chunk c of the item-major list is executed at time slot s with c = (s * stride) mod n_chunks
// (stride coprime to n_chunks, ~0.618 n_chunks), which spreads the ~200 chunks of a hot item over
// the whole epoch instead of running them all at once.
// ------------------------------------------------------------------------------------
// RMW (round 6; bit 0: P[u], bit 1: Q[j]): the row is updated by an sc1 load + an sc1 write-through STORE of the new value instead of
// an atomic delta -- no atomic unit involved, and an update of the same row that lands between this group's load and its store is
// LOST (racy Hogwild).  Measured (tools/probe_item_rmw.py, profiles/r06_item_rmw.json): with P[u] by RMW the Yelp2018-shape epoch
// takes 0.376 ms instead of 0.569 (0.64 of the roofline instead of 0.43) and the HBM-resident slice 15.3 ms instead of 20.9 (0.63
// instead of 0.46); Q[j] by RMW as well adds 1 %.  What it costs is a function of the COLLISION DENSITY c = groups in flight x
// sum_u p_u^2 (the expected number of other groups working on the same user while one holds it): c = 0.008 (650 k users): paired
// Recall@20 gaps equal to the atomic kernel's; c = 0.033 (160 k users): inside the bar, visibly worse (0.0010 vs 0.0001); c = 0.17
// (Yelp2018 shape): inside on the planted-community graph with a 10 % loss gap; lastfm under BPR.conf (1.9 k users): -0.019, far outside.
// So the host selects it by c (engine.resolve_p_update: c <= 0.01), never by shape; bit 1 (Q[j]) is kept for measurements only.
template <int LPR, int E, typename TAB, int RMW = 0>
__global__ __launch_bounds__(256) void bpr_hogwild_item_kernel(
    float *__restrict__ P, float *__restrict__ Q, int64_t p_bytes, int64_t q_bytes,
    const int32_t *__restrict__ u_idx, const int32_t *__restrict__ i_idx,
    const int32_t *__restrict__ j_idx, int64_t n, int chunk, int64_t n_chunks, int64_t chunk_stride,
    int64_t groups_active, int flush_every, HwRate rate, double *__restrict__ loss_out) {
    constexpr int GPW = kWave / LPR;
    float lr, cu, ci;
    if (!hw_rate_resolve(rate, lr, cu, ci)) return;
    __shared__ int32_t s_idx[4][GPW][3][kMaxChunk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + wave) * GPW + g;
    int32_t *su = s_idx[wave][g][0], *si = s_idx[wave][g][1], *sj = s_idx[wave][g][2];
    const TAB rsP = TAB::make(P, p_bytes), rsQ = TAB::make(Q, q_bytes);
    float loss = 0.f;
    double loss_acc = 0.0;

    for (int64_t slot = gid; slot < n_chunks && gid < groups_active; slot += groups_active) {
        const int64_t c = (int64_t)(((unsigned __int128)slot * (unsigned __int128)chunk_stride) % (unsigned __int128)n_chunks);
        const int64_t t0 = c * chunk;
        const int len = (int)((n - t0) < chunk ? (n - t0) : chunk);
        for (int k = r; k < len; k += LPR) { su[k] = u_idx[t0 + k]; si[k] = i_idx[t0 + k]; sj[k] = j_idx[t0 + k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        int cur_i = -1, since_flush = 0;
        Row<E> qi, qi0;
#pragma unroll
        for (int e = 0; e < E; e++) qi.v[e] = qi0.v[e] = 0.f;
        int ut = su[0], jt = sj[0];
        constexpr int LDP = (RMW & 1) ? LD_SC1 : LD_PLAIN, LDJ = (RMW & 2) ? LD_SC1 : LD_PLAIN;
        constexpr int UPP = (RMW & 1) ? UP_STORE_SC1 : UP_ATOMIC, UPJ = (RMW & 2) ? UP_STORE_SC1 : UP_ATOMIC;
        Row<E> pu = hw_load_row<LPR, E, LDP>(rsP, ut, r);
        Row<E> qj = hw_load_row<LPR, E, LDJ>(rsQ, jt, r);
        for (int k = 0; k < len; k++) {
            const int it = si[k];
            if (it != cur_i || since_flush >= flush_every) {
                if (cur_i >= 0) hw_update_row<LPR, E, UP_ATOMIC>(rsQ, cur_i, r, qi0, qi);
                qi = hw_load_row<LPR, E, LD_SC1>(rsQ, it, r);      // bypass L1: pick up other runs' flushes
                qi0 = qi; cur_i = it; since_flush = 0;
            }
            // The next triplet's rows go in flight under this one's math.  The 64-bit addressing flavour (TabPtr) issues them UNCONDITIONALLY
            // (the chunk's last triplet re-reads its own rows, unused): under `if (more)` its loads were if-converted into loads + selects,
            // the selects waited for the loads on the spot (s_waitcnt vmcnt(0) right behind the prefetch) and the software pipelining was
            // gone -- 168.9 instead of 120.5 ms on the 10 M-user HBM-resident slice with P[u] by load + store (round 6,
            // tools/probe_p_table_size.py).  The buffer-descriptor flavour keeps the branch: its loads stay behind a real skip, and the
            // unconditional form measured 11 % slower there (16.6 against 14.9 ms at 1.25 M users).
            const bool more = (k + 1 < len);
            int un = ut, jn = jt;
            Row<E> npu = pu, nqj = qj;
            if (TAB::kPrefetchAlways || more) {
                const int kn = more ? k + 1 : k;
                un = su[kn]; jn = sj[kn];
                npu = hw_load_row<LPR, E, LDP>(rsP, un, r);
                nqj = hw_load_row<LPR, E, LDJ>(rsQ, jn, r);
            }
            float di = 0.f, dj = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) { di += pu.v[e] * qi.v[e]; dj += pu.v[e] * qj.v[e]; }
            di = row_allreduce_sum<LPR>(di); dj = row_allreduce_sum<LPR>(dj);
            const float s = 1.0f / (1.0f + expf(-(di - dj)));
            const float gsc = lr * (1.0f - s);
            Row<E> pun, qjn;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float p1 = pu.v[e] + gsc * (qi.v[e] - qj.v[e]);
                float a = qi.v[e] + gsc * p1, b = qj.v[e] - gsc * p1;
                qi.v[e] = a - ci * a; qjn.v[e] = b - ci * b;
                pun.v[e] = p1 - cu * p1;
            }
            hw_update_row<LPR, E, UPP>(rsP, ut, r, pu, pun);
            hw_update_row<LPR, E, UPJ>(rsQ, jt, r, qj, qjn);
            loss += neg_log_sigmoid(di - dj);
            since_flush++;
            if (more) {   // rows this group just changed supersede the prefetched copy
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (un == ut) npu.v[e] = pun.v[e];
                    if (jn == jt) nqj.v[e] = qjn.v[e]; else if (jn == cur_i) nqj.v[e] = qi.v[e];
                }
            }
            pu = npu; qj = nqj; ut = un; jt = jn;
        }
        if (cur_i >= 0) hw_update_row<LPR, E, UP_ATOMIC>(rsQ, cur_i, r, qi0, qi);
        loss_acc += (double)loss; loss = 0.f;
        __builtin_amdgcn_wave_barrier();
    }
    if (r != 0) loss_acc = 0.0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) loss_acc += __shfl_xor(loss_acc, m, kWave);
    if (lane == 0 && loss_acc != 0.0) atomicAdd(loss_out, loss_acc);
}

template <int LPR, int E>
int launch_hogwild_item(float *P, float *Q, int64_t pb, int64_t qb, const int32_t *u, const int32_t *i,
                        const int32_t *j, int64_t n, int chunk, int64_t groups, int flush_every, HwRate rate,
                        double *loss, int variant, hipStream_t st) {
    constexpr int GPW = kWave / LPR;
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    const int64_t default_groups = (int64_t)256 * 4 * GPW, max_groups = (int64_t)256 * 8 * 4 * GPW;
    if (groups <= 0) groups = default_groups;
    if (groups > max_groups) groups = max_groups;
    if (groups > n_chunks) groups = n_chunks;
    // stride ~ 0.618 n_chunks, made coprime with n_chunks: consecutive time slots visit far-apart chunks
    int64_t stride = (int64_t)((double)n_chunks * 0.6180339887498949);
    if (stride < 1) stride = 1;
    auto gcd = [](int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; };
    while (gcd(stride, n_chunks) != 1) stride++;
    const unsigned blocks = (unsigned)((groups + 4 * GPW - 1) / (4 * GPW));
#define QREC_ITEM_LAUNCH(RMW)                                                                                                     \
    do {                                                                                                                          \
        if (pb < kBufLimit && qb < kBufLimit && !force_64bit_addressing())                                                        \
            hipLaunchKernelGGL((bpr_hogwild_item_kernel<LPR, E, TabBuf, RMW>), dim3(blocks), dim3(256), 0, st, P, Q, pb, qb, u, i, j, n, \
                               chunk, n_chunks, stride, groups, flush_every, rate, loss);                                         \
        else                                                                                                                      \
            hipLaunchKernelGGL((bpr_hogwild_item_kernel<LPR, E, TabPtr, RMW>), dim3(blocks), dim3(256), 0, st, P, Q, pb, qb, u, i, j, n, \
                               chunk, n_chunks, stride, groups, flush_every, rate, loss);                                         \
    } while (0)
    switch (variant) {
        case QREC_HW_DEFAULT:
        case QREC_HW_ATOMIC: QREC_ITEM_LAUNCH(0); break;
        case QREC_HW_P_RMW: QREC_ITEM_LAUNCH(1); break;
        case QREC_HW_PQ_RMW: QREC_ITEM_LAUNCH(3); break;
        default: set_error("qrec_bpr_sgd_hogwild_item_major: unknown variant %d", variant); return QREC_ERR_INVALID;
    }
#undef QREC_ITEM_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

template <int LPR, int E>
int launch_hogwild(float *P, float *Q, int64_t pb, int64_t qb, const int32_t *u,
                   const int32_t *i, const int32_t *j, int64_t n, int chunk, int64_t groups,
                   HwRate rate, double *loss, int variant, hipStream_t st) {
    constexpr int GPW = kWave / LPR;
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    // Groups in flight.  The kernel is bound by the L2 atomic units, not by latency: measured at
    // the Yelp2018 shape, 4,096 groups (one 256-thread block per CU) already run at the full
    // 0.70 ms/epoch, and every extra group in flight only adds read staleness on hot rows
    // (deviation from the sequential result 0.11% at 4k groups vs 0.35% at 32k).  So the
    // default is one block per CU; callers may ask for more (up to 8 blocks per CU) or fewer.
    const int64_t default_groups = (int64_t)256 * 4 * GPW;
    const int64_t max_groups = (int64_t)256 * 8 * 4 * GPW;
    if (groups <= 0) groups = default_groups;
    if (groups > max_groups) groups = max_groups;
    if (groups > n_chunks) groups = n_chunks;
    const unsigned blocks = (unsigned)((groups + 4 * GPW - 1) / (4 * GPW));
#define QREC_HW_LAUNCH(LOADP, UPD)                                                                          \
    do {                                                                                                    \
        if (pb < kBufLimit && qb < kBufLimit && !force_64bit_addressing())                                  \
            hipLaunchKernelGGL((bpr_hogwild_kernel<LPR, E, LOADP, UPD, TabBuf>), dim3(blocks), dim3(256), 0, st, \
                               P, Q, pb, qb, u, i, j, n, chunk, n_chunks, groups, rate, loss);              \
        else                                                                                                \
            hipLaunchKernelGGL((bpr_hogwild_kernel<LPR, E, LOADP, UPD, TabPtr>), dim3(blocks), dim3(256), 0, st, \
                               P, Q, pb, qb, u, i, j, n, chunk, n_chunks, groups, rate, loss);              \
    } while (0)
    switch (variant) {
        case QREC_HW_PLAIN_RMW: QREC_HW_LAUNCH(LD_PLAIN, UP_STORE); break;
        case QREC_HW_SC1_RMW: QREC_HW_LAUNCH(LD_SC1, UP_STORE_SC1); break;
        case QREC_HW_DEFAULT:
        case QREC_HW_ATOMIC: QREC_HW_LAUNCH(LD_PLAIN, UP_ATOMIC); break;
        case QREC_HW_SC1_ATOMIC: QREC_HW_LAUNCH(LD_SC1, UP_ATOMIC); break;
        default: set_error("qrec_bpr_sgd_hogwild: unknown variant %d", variant); return QREC_ERR_INVALID;
    }
#undef QREC_HW_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

template <typename T>
int launch_ordered(void *P, void *Q, int d, int ld, const int32_t *u, const int32_t *i,
                   const int32_t *j, int64_t n, double lr, double regU, double regI, double *loss,
                   hipStream_t st) {
    const T tlr = (T)lr, cu = (T)lr * (T)regU, ci = (T)lr * (T)regI;  // numpy: (lr*reg)*row
#define QREC_ORD_LAUNCH(EPL)                                                                   \
    hipLaunchKernelGGL((bpr_ordered_kernel<T, EPL>), dim3(1), dim3(64), 0, st, (T *)P, (T *)Q, d, \
                       ld, u, i, j, n, tlr, cu, ci, loss)
    if (d <= 64) QREC_ORD_LAUNCH(1);
    else if (d <= 128) QREC_ORD_LAUNCH(2);
    else QREC_ORD_LAUNCH(4);
#undef QREC_ORD_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

template <typename T, int VAR>
int launch_mf_ordered(void *P, void *Q, void *Bu, void *Bi, int d, int ld, const int32_t *u, const int32_t *i,
                      const double *r, int64_t n, double lr, double regU, double regI, double regB, double gmean,
                      double *loss, hipStream_t st) {
#define QREC_MF_LAUNCH(EPL)                                                                             \
    hipLaunchKernelGGL((mf_ordered_kernel<T, EPL, VAR>), dim3(1), dim3(64), 0, st, (T *)P, (T *)Q, (T *)Bu, \
                       (T *)Bi, d, ld, u, i, r, n, (T)lr, (T)regU, (T)regI, (T)regB, (T)gmean, loss)
    if (d <= 64) QREC_MF_LAUNCH(1);
    else if (d <= 128) QREC_MF_LAUNCH(2);
    else QREC_MF_LAUNCH(4);
#undef QREC_MF_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

template <typename T>
int dispatch_mf(int variant, void *P, void *Q, void *Bu, void *Bi, int d, int ld, const int32_t *u, const int32_t *i,
                const double *r, int64_t n, double lr, double regU, double regI, double regB, double gmean,
                double *loss, hipStream_t st) {
    switch (variant) {
        case 0: return launch_mf_ordered<T, 0>(P, Q, Bu, Bi, d, ld, u, i, r, n, lr, regU, regI, regB, gmean, loss, st);
        case 1: return launch_mf_ordered<T, 1>(P, Q, Bu, Bi, d, ld, u, i, r, n, lr, regU, regI, regB, gmean, loss, st);
        case 2: return launch_mf_ordered<T, 2>(P, Q, Bu, Bi, d, ld, u, i, r, n, lr, regU, regI, regB, gmean, loss, st);
        default: return launch_mf_ordered<T, 3>(P, Q, Bu, Bi, d, ld, u, i, r, n, lr, regU, regI, regB, gmean, loss, st);
    }
}

template <typename T>
int launch_sbpr(void *P, void *Q, const void *bias, int d, int ld, const int32_t *rows, int64_t n, double lr, double regU, double regI, double bb,
                const double *sums_in, double *loss2, hipStream_t st) {
    const T tlr = (T)lr, cu = (T)lr * (T)regU, ci = (T)lr * (T)regI;
#define QREC_SBPR_LAUNCH(EPL)                                                                                                      \
    hipLaunchKernelGGL((sbpr_ordered_kernel<T, EPL>), dim3(1), dim3(64), 0, st, (T *)P, (T *)Q, (const T *)bias, d, ld, rows, n, tlr, cu, ci, \
                       regU, regI, bb, sums_in, loss2)
    if (d <= 64) QREC_SBPR_LAUNCH(1);
    else if (d <= 128) QREC_SBPR_LAUNCH(2);
    else QREC_SBPR_LAUNCH(4);
#undef QREC_SBPR_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

template <typename T>
int launch_tbpr(void *P, void *Q, int d, int ld, const int32_t *u, const int32_t *a, const int32_t *b, int64_t n, double lr,
                double regU, double regI, const double *sums_in, double *loss, hipStream_t st) {
    const T tlr = (T)lr, cu = (T)lr * (T)regU, ci = (T)lr * (T)regI;
#define QREC_TBPR_LAUNCH(EPL)                                                                                          \
    hipLaunchKernelGGL((tbpr_ordered_kernel<T, EPL>), dim3(1), dim3(64), 0, st, (T *)P, (T *)Q, d, ld, u, a, b, n, tlr, cu, ci, \
                       regU, regI, sums_in, loss)
    if (d <= 64) QREC_TBPR_LAUNCH(1);
    else if (d <= 128) QREC_TBPR_LAUNCH(2);
    else QREC_TBPR_LAUNCH(4);
#undef QREC_TBPR_LAUNCH
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // namespace

extern "C" {

int qrec_bpr_sgd_ordered(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld,
                         const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int64_t n,
                         double lr, double regU, double regI, double *d_loss, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_loss && n >= 0, "qrec_bpr_sgd_ordered: null argument");
    QREC_REQUIRE(n == 0 || (d_u && d_i && d_j), "qrec_bpr_sgd_ordered: null index array");
    QREC_REQUIRE(d >= 1 && d <= 256 && ld >= d, "qrec_bpr_sgd_ordered: need 1 <= d <= 256, ld >= d (got d=%d ld=%d)", d, ld);
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_bpr_sgd_ordered: bad dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    if (n == 0) { QREC_HIP_CHECK(hipMemsetAsync(d_loss, 0, sizeof(double), st)); return QREC_OK; }
    return dtype == QREC_F64
               ? launch_ordered<double>(d_P, d_Q, d, ld, d_u, d_i, d_j, n, lr, regU, regI, d_loss, st)
               : launch_ordered<float>(d_P, d_Q, d, ld, d_u, d_i, d_j, n, lr, regU, regI, d_loss, st);
}

int qrec_sbpr_sgd_ordered(void *d_P, void *d_Q, const void *d_bias, int dtype, int32_t d, int32_t ld, const int32_t *d_rows, int64_t n, double lr,
                          double regU, double regI, double bias_sumsq, const double *d_sums_in, double *d_loss2, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_bias && d_sums_in && d_loss2 && n >= 0, "qrec_sbpr_sgd_ordered: null argument");
    QREC_REQUIRE(n == 0 || d_rows, "qrec_sbpr_sgd_ordered: null row array");
    QREC_REQUIRE(d >= 1 && d <= 256 && ld >= d, "qrec_sbpr_sgd_ordered: need 1 <= d <= 256, ld >= d (got d=%d ld=%d)", d, ld);
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_sbpr_sgd_ordered: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    return dtype == QREC_F64 ? launch_sbpr<double>(d_P, d_Q, d_bias, d, ld, d_rows, n, lr, regU, regI, bias_sumsq, d_sums_in, d_loss2, st)
                             : launch_sbpr<float>(d_P, d_Q, d_bias, d, ld, d_rows, n, lr, regU, regI, bias_sumsq, d_sums_in, d_loss2, st);
}

int qrec_tbpr_sgd_ordered(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld, const int32_t *d_u, const int32_t *d_a,
                          const int32_t *d_b, int64_t n, double lr, double regU, double regI, const double *d_sums_in,
                          double *d_loss2, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_sums_in && d_loss2 && n >= 0, "qrec_tbpr_sgd_ordered: null argument");
    QREC_REQUIRE(n == 0 || (d_u && d_a && d_b), "qrec_tbpr_sgd_ordered: null index array");
    QREC_REQUIRE(d >= 1 && d <= 256 && ld >= d, "qrec_tbpr_sgd_ordered: need 1 <= d <= 256, ld >= d (got d=%d ld=%d)", d, ld);
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_tbpr_sgd_ordered: bad dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    if (n == 0) { QREC_HIP_CHECK(hipMemsetAsync(d_loss2, 0, 2 * sizeof(double), st)); return QREC_OK; }
    return dtype == QREC_F64 ? launch_tbpr<double>(d_P, d_Q, d, ld, d_u, d_a, d_b, n, lr, regU, regI, d_sums_in, d_loss2, st)
                             : launch_tbpr<float>(d_P, d_Q, d, ld, d_u, d_a, d_b, n, lr, regU, regI, d_sums_in, d_loss2, st);
}

int qrec_mf_sgd_ordered(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld,
                        const int32_t *d_u, const int32_t *d_i, const double *d_rating, int64_t n,
                        double lr, double *d_loss, int variant, double regU, double regI, void *d_Bu,
                        void *d_Bi, double regB, double global_mean, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_loss && n >= 0, "qrec_mf_sgd_ordered: null argument");
    QREC_REQUIRE(n == 0 || (d_u && d_i && d_rating), "qrec_mf_sgd_ordered: null index array");
    QREC_REQUIRE(d >= 1 && d <= 256 && ld >= d, "qrec_mf_sgd_ordered: need 1 <= d <= 256, ld >= d");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_mf_sgd_ordered: bad dtype %d", dtype);
    QREC_REQUIRE(variant >= 0 && variant <= 3, "qrec_mf_sgd_ordered: variant must be 0 (BasicMF), 1 (PMF), 2 (SVD) or 3 (EE)");
    QREC_REQUIRE(variant < 2 || (d_Bu && d_Bi), "qrec_mf_sgd_ordered: SVD and EE need the bias vectors");
    hipStream_t st = as_stream(stream);
    if (n == 0) { QREC_HIP_CHECK(hipMemsetAsync(d_loss, 0, sizeof(double), st)); return QREC_OK; }
    return dtype == QREC_F64 ? dispatch_mf<double>(variant, d_P, d_Q, d_Bu, d_Bi, d, ld, d_u, d_i, d_rating, n, lr, regU, regI, regB, global_mean, d_loss, st)
                             : dispatch_mf<float>(variant, d_P, d_Q, d_Bu, d_Bi, d, ld, d_u, d_i, d_rating, n, lr, regU, regI, regB, global_mean, d_loss, st);
}

int qrec_svdpp_sgd_ordered(void *d_P, void *d_Q, void *d_Y, void *d_Bu, void *d_Bi, int dtype, int32_t d, int32_t ld,
                           const int64_t *d_rated_indptr, const int32_t *d_rated_items, const int32_t *d_u,
                           const int32_t *d_i, const double *d_rating, int64_t n, double lr, double regU, double regI,
                           double regB, double regY, double global_mean, double *d_loss, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_Y && d_Bu && d_Bi && d_rated_indptr && d_loss && n >= 0, "qrec_svdpp_sgd_ordered: null argument");
    QREC_REQUIRE(n == 0 || (d_u && d_i && d_rating && d_rated_items), "qrec_svdpp_sgd_ordered: null index array");
    QREC_REQUIRE(d >= 1 && d <= 256 && ld >= d, "qrec_svdpp_sgd_ordered: need 1 <= d <= 256, ld >= d");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_svdpp_sgd_ordered: bad dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    if (n == 0) { QREC_HIP_CHECK(hipMemsetAsync(d_loss, 0, sizeof(double), st)); return QREC_OK; }
#define QREC_SVDPP(T, EPL)                                                                                              \
    hipLaunchKernelGGL((svdpp_ordered_kernel<T, EPL>), dim3(1), dim3(64), 0, st, (T *)d_P, (T *)d_Q, (T *)d_Y, (T *)d_Bu, \
                       (T *)d_Bi, d, ld, d_rated_indptr, d_rated_items, d_u, d_i, d_rating, n, (T)lr, (T)regU, (T)regI,    \
                       (T)regB, (T)regY, (T)global_mean, d_loss)
    if (dtype == QREC_F64) { if (d <= 64) QREC_SVDPP(double, 1); else if (d <= 128) QREC_SVDPP(double, 2); else QREC_SVDPP(double, 4); }
    else { if (d <= 64) QREC_SVDPP(float, 1); else if (d <= 128) QREC_SVDPP(float, 2); else QREC_SVDPP(float, 4); }
#undef QREC_SVDPP
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_bpr_sgd_hogwild(float *d_P, float *d_Q, int64_t n_users, int64_t n_items, int32_t d, int32_t ld, const int32_t *d_u,
                         const int32_t *d_i, const int32_t *d_j, int64_t n, int32_t chunk,
                         int32_t grid_groups, float lr, float regU, float regI, double *d_loss,
                         int variant, const double *d_driver_state, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_loss && n >= 0, "qrec_bpr_sgd_hogwild: null argument");
    QREC_REQUIRE(n == 0 || (d_u && d_i && d_j), "qrec_bpr_sgd_hogwild: null index array");
    QREC_REQUIRE(ld >= d && d >= 1, "qrec_bpr_sgd_hogwild: need ld >= d >= 1");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256,
                 "qrec_bpr_sgd_hogwild: row stride must be 32, 64, 128 or 256 floats (pad d=%d up; got ld=%d)", d, ld);
    QREC_REQUIRE(chunk >= 1 && chunk <= kMaxChunk, "qrec_bpr_sgd_hogwild: chunk must be in 1..%d", kMaxChunk);
    if (n == 0) return QREC_OK;
    // tables below 4 GiB go through a buffer descriptor sized to the table (ids >= rows are dropped by the hardware
    // bounds check); larger ones through 64-bit addressing
    QREC_REQUIRE(n_users >= 1 && n_items >= 1, "qrec_bpr_sgd_hogwild: table row counts must be given");
    const int64_t full_p = n_users * (int64_t)ld * 4, full_q = n_items * (int64_t)ld * 4;
    hipStream_t st = as_stream(stream);
    const HwRate rate{lr, regU, regI, d_driver_state};
    switch (ld) {
        case 32: return launch_hogwild<16, 2>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, rate, d_loss, variant, st);
        case 64: return launch_hogwild<16, 4>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, rate, d_loss, variant, st);
        case 128: return launch_hogwild<32, 4>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, rate, d_loss, variant, st);
        default: return launch_hogwild<64, 4>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, rate, d_loss, variant, st);
    }
}

int qrec_bpr_sgd_hogwild_item_major(float *d_P, float *d_Q, int64_t n_users, int64_t n_items, int32_t d, int32_t ld,
                                    const int32_t *d_u,
                                    const int32_t *d_i, const int32_t *d_j, int64_t n, int32_t chunk,
                                    int32_t grid_groups, int32_t flush_every, float lr, float regU, float regI,
                                    double *d_loss, int variant, const double *d_driver_state, void *stream) {
    QREC_REQUIRE(d_P && d_Q && d_loss && n >= 0, "qrec_bpr_sgd_hogwild_item_major: null argument");
    QREC_REQUIRE(n == 0 || (d_u && d_i && d_j), "qrec_bpr_sgd_hogwild_item_major: null index array");
    QREC_REQUIRE(ld >= d && d >= 1, "qrec_bpr_sgd_hogwild_item_major: need ld >= d >= 1");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256,
                 "qrec_bpr_sgd_hogwild_item_major: row stride must be 32, 64, 128 or 256 floats (got ld=%d)", ld);
    QREC_REQUIRE(chunk >= 1 && chunk <= kMaxChunk && flush_every >= 1, "qrec_bpr_sgd_hogwild_item_major: bad chunk / flush interval");
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(n_users >= 1 && n_items >= 1, "qrec_bpr_sgd_hogwild_item_major: table row counts must be given");
    const int64_t full_p = n_users * (int64_t)ld * 4, full_q = n_items * (int64_t)ld * 4;
    hipStream_t st = as_stream(stream);
    const HwRate rate{lr, regU, regI, d_driver_state};
    switch (ld) {
        case 32: return launch_hogwild_item<16, 2>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, flush_every, rate, d_loss, variant, st);
        case 64: return launch_hogwild_item<16, 4>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, flush_every, rate, d_loss, variant, st);
        case 128: return launch_hogwild_item<32, 4>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, flush_every, rate, d_loss, variant, st);
        default: return launch_hogwild_item<64, 4>(d_P, d_Q, full_p, full_q, d_u, d_i, d_j, n, chunk, grid_groups, flush_every, rate, d_loss, variant, st);
    }
}

}  // extern "C"
