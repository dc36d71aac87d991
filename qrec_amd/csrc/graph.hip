// Graph-recommender kernels (LightGCN family): the TF-1.14 ops the reference strings together in
// model/ranking/LightGCN.py:11-41, as hand-written gfx950 kernels.
//
//   spmm_kernel            tf.sparse_tensor_dense_matmul(norm_adj, E)         LightGCN.py:17
//   bpr_batch_kernel       embedding_lookup x3 + bpr_loss + batch l2_loss and their gradients
//                          (scatter-add of the lookups' backward)  LightGCN.py:22-30, util/loss.py:3-6
//   adam_kernel            tf.train.AdamOptimizer.apply_gradients (dense)        LightGCN.py:31-32
//
// All HBM/L2-bandwidth bound (<= 0.5 FLOP/B): no MFMA here by design.
#include "common.h"

using namespace qrec;

namespace {

// ---------------------------------------------------------------------------------------
// CSR SpMM  Y = A X  (+ addend_scale * addend) ; optionally accum += Y.
// One group of LPR lanes (float4 per lane -> one whole row of ld = 4*LPR floats per group per
// load) walks one SEGMENT: a row, or a <= seg_len slice of a long row (the adjacency of an
// implicit-feedback graph is heavy-tailed: 6k-neighbour items next to 35-neighbour averages).
// A slice writes its partial sum to a scratch slot; a second tiny kernel adds a long row's
// partials in slice order, so the result is deterministic (no float atomics).
// (col, val) pairs are fetched LPR at a time with one coalesced load per lane and broadcast
// inside the group; the gathers of X rows are independent and issued 4 deep.
// Products and sums are not contracted: the scipy/TF CPU kernels round a*x and the add apart.
// Algorithmic bytes: nnz*(4+4) + 8*(rows+1) + 2*rows*d*4; the gathered operand
// (nnz*d*4 B of L2/MALL traffic) is reported separately.
// ---------------------------------------------------------------------------------------
template <int LPR>
__device__ inline f32x4 ld_row4(const float *__restrict__ X, int row, int r) {
    return *reinterpret_cast<const f32x4 *>(X + (int64_t)row * (4 * LPR) + 4 * r);
}

template <int LPR>
__device__ inline void spmm_epilogue(f32x4 acc, int row, int r, float *__restrict__ Y,
                                     const float *__restrict__ addend, float addend_scale,
                                     float *__restrict__ accum, const float *__restrict__ accum_init,
                                     const uint32_t *__restrict__ addend_row_mask = nullptr) {
#pragma clang fp contract(off)
    const int64_t off = (int64_t)row * (4 * LPR) + 4 * r;
    // addend_row_mask: the addend is only defined (and only non-zero) at the marked rows -- the batch gradient of a training step
    if (addend && (!addend_row_mask || ((addend_row_mask[row >> 5] >> (row & 31)) & 1u))) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(addend + off);
        acc = acc + addend_scale * a;
    }
    *reinterpret_cast<f32x4 *>(Y + off) = acc;
    if (accum) {      // accum = (accum_init ? accum_init : accum) + Y: the first layer starts the layer sum from the input table
        f32x4 s = *reinterpret_cast<const f32x4 *>((accum_init ? accum_init : accum) + off);
        s = s + acc;
        *reinterpret_cast<f32x4 *>(accum + off) = s;
    }
}

// broadcast of lane N of every 16-lane row to the row (DPP row_newbcast: VALU rate, no LDS crossbar)
template <int N>
__device__ __forceinline__ int row16_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false); }
template <int N>
__device__ __forceinline__ float row16_bcast(float v) { return __builtin_bit_cast(float, row16_bcast<N>(__builtin_bit_cast(int, v))); }

// (round 5, measured and removed: recomputing every value in the kernel as fl32(dinv[r]) * dinv[c] from the 279 KB degree vector instead
// of streaming the value array -- the same bits, 72.4 us against 70.3: the 4 B per non-zero of a coalesced stream become a gathered 4 B per
// non-zero, one more L2 transaction next to the two of the operand row it scales.  DESIGN.md s5.)
template <int LPR>
__global__ __launch_bounds__(256) void spmm_kernel(
    const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_beg,
    const int32_t *__restrict__ seg_len, const int32_t *__restrict__ seg_slot, int64_t n_segs,
    const int32_t *__restrict__ indices, const float *__restrict__ values,
    const float *__restrict__ X, float *__restrict__ Y, float *__restrict__ partial,
    const float *__restrict__ addend, float addend_scale, float *__restrict__ accum, const float *__restrict__ accum_init,
    const uint32_t *__restrict__ x_row_mask, const uint32_t *__restrict__ y_row_mask, const uint32_t *__restrict__ addend_row_mask) {
#pragma clang fp contract(off)
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    const int64_t n_groups = (int64_t)gridDim.x * 4 * GPW;
    for (int64_t s = gid; s < n_segs; s += n_groups) {
        if (y_row_mask) {   // only the marked output rows are wanted (last propagation layer: the batch's rows)
            const int row = seg_row[s];
            if (!((y_row_mask[row >> 5] >> (row & 31)) & 1u)) continue;
        }
        const int64_t beg = seg_beg[s];
        const int len = seg_len[s];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e0 = 0; e0 < len; e0 += LPR) {
            const int mine = e0 + r;
            int my_c = mine < len ? indices[beg + mine] : 0;
            const float my_v = mine < len ? values[beg + mine] : 0.f;
            const int cnt = (len - e0) < LPR ? (len - e0) : LPR;
            int k = 0;
            if constexpr (LPR == 16) {
                // ld = 64: a group IS a 16-lane DPP row, so (col, val) of entry K reach the group by row_newbcast:K -- eight
                // ds_bpermute per four non-zeros become eight VALU moves.  Entries past the segment's end carry value 0 and
                // the column of the batch's first entry (a row gathered anyway; acc + 0*x = acc), so the gathers always go
                // out four at a time.
                if (mine >= len) my_c = row16_bcast<0>(my_c);
                if (x_row_mask && !((x_row_mask[my_c >> 5] >> (my_c & 31)) & 1u)) my_c = -1;      // exactly-zero operand row: not fetched
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#define QREC_QUAD(K)                                                                                                     \
                if (K < cnt) {                                                                                           \
                    const int c0 = row16_bcast<K>(my_c), c1 = row16_bcast<K + 1>(my_c), c2 = row16_bcast<K + 2>(my_c), c3 = row16_bcast<K + 3>(my_c); \
                    const float v0 = row16_bcast<K>(my_v), v1 = row16_bcast<K + 1>(my_v), v2 = row16_bcast<K + 2>(my_v), v3 = row16_bcast<K + 3>(my_v); \
                    const f32x4 x0 = c0 >= 0 ? ld_row4<LPR>(X, c0, r) : zero, x1 = c1 >= 0 ? ld_row4<LPR>(X, c1, r) : zero;       \
                    const f32x4 x2 = c2 >= 0 ? ld_row4<LPR>(X, c2, r) : zero, x3 = c3 >= 0 ? ld_row4<LPR>(X, c3, r) : zero;       \
                    acc = acc + v0 * x0; acc = acc + v1 * x1; acc = acc + v2 * x2; acc = acc + v3 * x3;                   \
                }
                QREC_QUAD(0) QREC_QUAD(4) QREC_QUAD(8) QREC_QUAD(12)
#undef QREC_QUAD
                continue;
            }
            if (x_row_mask) {
                // sparse operand (first backward SpMM: X = the batch gradient, <= 3B non-zero rows): a
                // clear mask bit means the row is exactly zero; it is not fetched and contributes +0
                // (bit-identical result, ~90% fewer row gathers)
                if (mine < len && !((x_row_mask[my_c >> 5] >> (my_c & 31)) & 1u)) my_c = -1;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                for (; k + 4 <= cnt; k += 4) {
                    const int c0 = __shfl(my_c, g * LPR + k, kWave), c1 = __shfl(my_c, g * LPR + k + 1, kWave);
                    const int c2 = __shfl(my_c, g * LPR + k + 2, kWave), c3 = __shfl(my_c, g * LPR + k + 3, kWave);
                    const float v0 = __shfl(my_v, g * LPR + k, kWave), v1 = __shfl(my_v, g * LPR + k + 1, kWave);
                    const float v2 = __shfl(my_v, g * LPR + k + 2, kWave), v3 = __shfl(my_v, g * LPR + k + 3, kWave);
                    const f32x4 x0 = c0 >= 0 ? ld_row4<LPR>(X, c0, r) : zero, x1 = c1 >= 0 ? ld_row4<LPR>(X, c1, r) : zero;
                    const f32x4 x2 = c2 >= 0 ? ld_row4<LPR>(X, c2, r) : zero, x3 = c3 >= 0 ? ld_row4<LPR>(X, c3, r) : zero;
                    acc = acc + v0 * x0; acc = acc + v1 * x1; acc = acc + v2 * x2; acc = acc + v3 * x3;
                }
                for (; k < cnt; k++) {
                    const int c = __shfl(my_c, g * LPR + k, kWave);
                    const float v = __shfl(my_v, g * LPR + k, kWave);
                    acc = acc + v * (c >= 0 ? ld_row4<LPR>(X, c, r) : zero);
                }
                continue;
            }
            for (; k + 4 <= cnt; k += 4) {
                const int c0 = __shfl(my_c, g * LPR + k, kWave), c1 = __shfl(my_c, g * LPR + k + 1, kWave);
                const int c2 = __shfl(my_c, g * LPR + k + 2, kWave), c3 = __shfl(my_c, g * LPR + k + 3, kWave);
                const float v0 = __shfl(my_v, g * LPR + k, kWave), v1 = __shfl(my_v, g * LPR + k + 1, kWave);
                const float v2 = __shfl(my_v, g * LPR + k + 2, kWave), v3 = __shfl(my_v, g * LPR + k + 3, kWave);
                const f32x4 x0 = ld_row4<LPR>(X, c0, r), x1 = ld_row4<LPR>(X, c1, r);
                const f32x4 x2 = ld_row4<LPR>(X, c2, r), x3 = ld_row4<LPR>(X, c3, r);
                acc = acc + v0 * x0; acc = acc + v1 * x1; acc = acc + v2 * x2; acc = acc + v3 * x3;
            }
            for (; k < cnt; k++) {
                const int c = __shfl(my_c, g * LPR + k, kWave);
                const float v = __shfl(my_v, g * LPR + k, kWave);
                acc = acc + v * ld_row4<LPR>(X, c, r);
            }
        }
        const int slot = seg_slot[s];
        if (slot < 0) spmm_epilogue<LPR>(acc, seg_row[s], r, Y, addend, addend_scale, accum, accum_init, addend_row_mask);
        else *reinterpret_cast<f32x4 *>(partial + (int64_t)slot * (4 * LPR) + 4 * r) = acc;
    }
}

// long rows: Y[row] = sum of the row's partial slots, then the epilogue.  One WAVEFRONT per long row: the
// row's slots are dealt round-robin to the wavefront's GPW groups (8 independent loads in flight each), each
// group adds its slots in slice order, and the group sums are folded in a fixed butterfly -- deterministic,
// and the dependent-load chain of a 6k-neighbour row (49 slots) is 2 rounds instead of 7.
template <int LPR>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int32_t *__restrict__ long_row,
                                                         const int32_t *__restrict__ long_first,
                                                         const int32_t *__restrict__ long_count, int n_long,
                                                         const float *__restrict__ partial, float *__restrict__ Y,
                                                         const float *__restrict__ addend, float addend_scale,
                                                         float *__restrict__ accum, const float *__restrict__ accum_init,
                                                         const uint32_t *__restrict__ y_row_mask, const uint32_t *__restrict__ addend_row_mask) {
#pragma clang fp contract(off)
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= n_long) return;
    if (y_row_mask) {
        const int row = long_row[wid];
        if (!((y_row_mask[row >> 5] >> (row & 31)) & 1u)) return;
    }
    const int first = long_first[wid], cnt = long_count[wid];
    const float *p = partial + (int64_t)first * (4 * LPR) + 4 * r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = g;
    for (; k + 7 * GPW < cnt; k += 8 * GPW) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const f32x4 *>(p + (int64_t)(k + q * GPW) * (4 * LPR));
#pragma unroll
        for (int q = 0; q < 8; q++) acc = acc + v[q];
    }
    for (; k < cnt; k += GPW) acc = acc + *reinterpret_cast<const f32x4 *>(p + (int64_t)k * (4 * LPR));
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
    }
    if (g == 0) spmm_epilogue<LPR>(acc, long_row[wid], r, Y, addend, addend_scale, accum, accum_init, addend_row_mask);
}

// Bitmap of the rows a batch touches: users u, items n_users+i and n_users+j (the caller clears it first).  These
// are the only rows of the LAST propagation layer anybody reads (embedding_lookup, LightGCN.py:22-24) and the
// only non-zero rows of the batch gradient (first backward SpMM).
__global__ void mark_batch_rows_kernel(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                       const int32_t *__restrict__ j, int B, int n_users, uint32_t *__restrict__ mask) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int ru = u[b], ri = n_users + i[b], rj = n_users + j[b];
    atomicOr(mask + (ru >> 5), 1u << (ru & 31));
    atomicOr(mask + (ri >> 5), 1u << (ri & 31));
    atomicOr(mask + (rj >> 5), 1u << (rj & 31));
}

// ---------------------------------------------------------------------------------------
// Batch BPR loss + gradients.  Per triplet b (a group of LPR lanes; lane r holds columns
// r, r+LPR, ... so that every atomic covers one contiguous 4*LPR-byte segment -- see
// bpr_sgd.hip):
//   ub = S[u]/div, ib = S[nu+i]/div, jb = S[nu+j]/div            (mean over layers folded in)
//   score = ub.ib - ub.jb ; s = sigmoid(score)
//   loss += -log(s + eps) + reg/2 * (|ub|^2 + |ib|^2 + |jb|^2)
//   g = -s(1-s)/(s+eps)
//   dE[u] += g (ib - jb) + reg ub ;  dE[nu+i] += g ub + reg ib ;  dE[nu+j] += -g ub + reg jb
// Bytes: 3 rows read + 3 rows of atomics per triplet.
// ORDERED (parity mode): the three row gradients go to slots b, B + b, 2B + b of an ordered-scatter workspace instead
// (contrib / keys; ordered.hip adds them row by row in slot order, one lookup at a time): same bits on every launch.
// ---------------------------------------------------------------------------------------
template <int LPR, int E, bool ORDERED>
__global__ __launch_bounds__(256) void bpr_batch_kernel(
    const float *__restrict__ S, float div, int n_users, const int32_t *__restrict__ u_idx,
    const int32_t *__restrict__ i_idx, const int32_t *__restrict__ j_idx, int B, float eps, float reg,
    float *__restrict__ dE, uint32_t de_bytes, double *__restrict__ loss_out, uint32_t *__restrict__ row_mask,
    float *__restrict__ contrib, int32_t *__restrict__ keys) {
    constexpr int GPW = kWave / LPR;
    constexpr int LD = LPR * E;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    const int64_t n_groups = (int64_t)gridDim.x * 4 * GPW;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(dE, de_bytes);
    double loss = 0.0;
    for (int64_t b = gid; b < B; b += n_groups) {
        const int ru = u_idx[b], ri = n_users + i_idx[b], rj = n_users + j_idx[b];
        float ub[E], ib[E], jb[E];
        float di = 0.f, dj = 0.f, sq = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) {
            ub[e] = S[(int64_t)ru * LD + r + LPR * e] / div;
            ib[e] = S[(int64_t)ri * LD + r + LPR * e] / div;
            jb[e] = S[(int64_t)rj * LD + r + LPR * e] / div;
            di += ub[e] * ib[e]; dj += ub[e] * jb[e];
            sq += ub[e] * ub[e] + ib[e] * ib[e] + jb[e] * jb[e];
        }
        di = row_allreduce_sum<LPR>(di); dj = row_allreduce_sum<LPR>(dj); sq = row_allreduce_sum<LPR>(sq);
        const float s = 1.0f / (1.0f + expf(-(di - dj)));
        const float gsc = -(s * (1.0f - s)) / (s + eps);
        if (r == 0) {
            loss += (double)(-logf(s + eps)) + 0.5 * (double)reg * (double)sq;
            if (row_mask) {   // rows of dE that become non-zero (for the sparse-operand SpMM)
                atomicOr(row_mask + (ru >> 5), 1u << (ru & 31)); atomicOr(row_mask + (ri >> 5), 1u << (ri & 31));
                atomicOr(row_mask + (rj >> 5), 1u << (rj & 31));
            }
        }
        if constexpr (ORDERED) {
#pragma clang fp contract(off)
            if (r == 0) { keys[b] = ru; keys[(int64_t)B + b] = ri; keys[2 * (int64_t)B + b] = rj; }
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int col = r + LPR * e;
                contrib[b * LD + col] = gsc * (ib[e] - jb[e]) + reg * ub[e];
                contrib[((int64_t)B + b) * LD + col] = gsc * ub[e] + reg * ib[e];
                contrib[(2 * (int64_t)B + b) * LD + col] = -gsc * ub[e] + reg * jb[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t col = (uint32_t)(r + LPR * e) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(gsc * (ib[e] - jb[e]) + reg * ub[e], rs, (int)((uint32_t)ru * LD * 4u + col), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(gsc * ub[e] + reg * ib[e], rs, (int)((uint32_t)ri * LD * 4u + col), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(-gsc * ub[e] + reg * jb[e], rs, (int)((uint32_t)rj * LD * 4u + col), 0, 0);
            }
        }
    }
    // one fp64 atomic per BLOCK: same-address atomics are serialised in L2 (~10 ns each); one per wavefront was
    // two thirds of this kernel's 30 us at ld = 256 (a wavefront per triplet, 2,048 of them)
    __shared__ double s_loss[4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) loss += __shfl_xor(loss, m, kWave);
    if (lane == 0) s_loss[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        if (t != 0.0) atomicAdd(loss_out, t);
    }
}

// ---------------------------------------------------------------------------------------
// Dense Adam in the form of TF 1.14's ApplyAdam functor (training_ops.cc), all fp32,
// g = gscale*G + l2*theta (l2 != 0 folds in the gradient of reg*tf.nn.l2_loss(theta), BPR.py:83):
//   m += (g - m) * (1 - beta1) ; v += (g*g - v) * (1 - beta2) ; theta -= (m * alpha) / (sqrt(v) + eps)
// alpha = lr*sqrt(1-beta2_power)/(1-beta1_power) comes from the host (fp32 beta powers, as TF keeps
// them).  Streams theta, m, v, G in and theta, m, v out: 7 * 4 B per element.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ theta, float *__restrict__ m,
                                                   float *__restrict__ v, const float *__restrict__ G,
                                                   int64_t n4, float gscale, float l2, float alpha, float b1,
                                                   float b2, float eps) {
#pragma clang fp contract(off)
    f32x4 *t4 = reinterpret_cast<f32x4 *>(theta), *m4 = reinterpret_cast<f32x4 *>(m), *v4 = reinterpret_cast<f32x4 *>(v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(G);
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        f32x4 mm = m4[k], vv = v4[k], th = t4[k];
        const f32x4 g = gscale * g4[k] + l2 * th;
        mm = mm + (g - mm) * omb1;
        vv = vv + (g * g - vv) * omb2;
        th.x -= (mm.x * alpha) / (sqrtf(vv.x) + eps); th.y -= (mm.y * alpha) / (sqrtf(vv.y) + eps);
        th.z -= (mm.z * alpha) / (sqrtf(vv.z) + eps); th.w -= (mm.w * alpha) / (sqrtf(vv.w) + eps);
        m4[k] = mm; v4[k] = vv; t4[k] = th;
    }
}

template <int LPR>
int launch_spmm(const int32_t *seg_row, const int64_t *seg_beg, const int32_t *seg_len, const int32_t *seg_slot,
                int64_t n_segs, const int32_t *long_row, const int32_t *long_first, const int32_t *long_count,
                int n_long, const int32_t *indices, const float *values, const float *X, float *Y, float *partial,
                const float *addend, float addend_scale, float *accum, const float *accum_init, const uint32_t *x_row_mask,
                const uint32_t *y_row_mask, const uint32_t *addend_row_mask, hipStream_t st) {
    constexpr int GPW = kWave / LPR;
    int64_t blocks = (n_segs + 4 * GPW - 1) / (4 * GPW);
    if (blocks > 256 * 8) blocks = 256 * 8;
    blocks = (blocks + 7) & ~(int64_t)7;     // a multiple of the 8 XCDs: segment p always lands on XCD (p / groups per block) % 8
    hipLaunchKernelGGL((spmm_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, seg_row, seg_beg, seg_len,
                       seg_slot, n_segs, indices, values, X, Y, partial, addend, addend_scale, accum, accum_init, x_row_mask, y_row_mask, addend_row_mask);
    QREC_LAUNCH_CHECK();
    if (n_long > 0) {
        hipLaunchKernelGGL((spmm_fixup_kernel<LPR>), dim3((unsigned)((n_long + 3) / 4)), dim3(256),
                           0, st, long_row, long_first, long_count, n_long, partial, Y, addend, addend_scale, accum, accum_init, y_row_mask, addend_row_mask);
        QREC_LAUNCH_CHECK();
    }
    return QREC_OK;
}

// ---------------------------------------------------------------------------------------
// Per-layer l2-normalised propagation (SEPT.py:142-160: every propagated layer enters the view's sum as
// tf.math.l2_normalize(x, axis=1) = x * rsqrt(max(sum x^2, 1e-12))), forward and backward.  LPR lanes per row.
//   l2norm_accum_kernel   inv[row] = rsqrt(max(|x|^2, 1e-12));  S[row] += x * inv
//   l2norm_bwd_kernel     out[row] = (dS - z (z.dS)) * inv,  z = x * inv     (gradient w.r.t. x of the term above)
//   scale_copy_kernel     dst = alpha * src                                   (the Variable / 2 of SEPT.py:129-130)
// ---------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void l2norm_accum_kernel(const float *__restrict__ X, int64_t n_rows, float *__restrict__ S,
                                                           float *__restrict__ inv) {
#pragma clang fp contract(off)
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n_rows; row += (int64_t)gridDim.x * 4 * GPW) {
        const int64_t off = row * (4 * LPR) + 4 * r;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(X + off);
        float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        ss = row_allreduce_sum<LPR>(ss);
        const float iv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        f32x4 s = *reinterpret_cast<const f32x4 *>(S + off);
        s = s + x * iv;
        *reinterpret_cast<f32x4 *>(S + off) = s;
        if (r == 0) inv[row] = iv;
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ X, const float *__restrict__ inv,
                                                         const float *__restrict__ dS, int64_t n_rows, float *__restrict__ out) {
#pragma clang fp contract(off)
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n_rows; row += (int64_t)gridDim.x * 4 * GPW) {
        const int64_t off = row * (4 * LPR) + 4 * r;
        const float iv = inv[row];
        const f32x4 z = *reinterpret_cast<const f32x4 *>(X + off) * iv;
        const f32x4 d = *reinterpret_cast<const f32x4 *>(dS + off);
        float dot = z.x * d.x + z.y * d.y + z.z * d.z + z.w * d.w;
        dot = row_allreduce_sum<LPR>(dot);
        *reinterpret_cast<f32x4 *>(out + off) = (d - z * dot) * iv;
    }
}

__global__ __launch_bounds__(256) void scale_copy_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n4, float alpha) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4 *>(dst)[k] = alpha * reinterpret_cast<const f32x4 *>(src)[k];
}

}  // namespace

extern "C" {

int qrec_spmm_csr(const int32_t *d_seg_row, const int64_t *d_seg_beg, const int32_t *d_seg_len,
                  const int32_t *d_seg_slot, int64_t n_segs, const int32_t *d_long_row,
                  const int32_t *d_long_first, const int32_t *d_long_count, int32_t n_long,
                  const int32_t *d_indices, const float *d_values, const float *d_X, float *d_Y,
                  float *d_partial, int32_t ld, const float *d_addend, float addend_scale, float *d_accum,
                  const float *d_accum_init, const uint32_t *d_x_row_mask, const uint32_t *d_y_row_mask,
                  const uint32_t *d_addend_row_mask, void *stream) {
    QREC_REQUIRE(d_seg_row && d_seg_beg && d_seg_len && d_seg_slot && d_indices && d_values && d_X && d_Y,
                 "qrec_spmm_csr: null argument");
    QREC_REQUIRE(n_long == 0 || (d_long_row && d_long_first && d_long_count && d_partial), "qrec_spmm_csr: long-row plan incomplete");
    QREC_REQUIRE(d_X != d_Y, "qrec_spmm_csr: in-place SpMM is not supported (Y may alias addend, not X)");
    QREC_REQUIRE(!d_accum_init || d_accum, "qrec_spmm_csr: d_accum_init without d_accum");
    QREC_REQUIRE(!d_addend_row_mask || d_addend, "qrec_spmm_csr: d_addend_row_mask without d_addend");
    if (n_segs == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
#define QREC_SPMM(LPR) return launch_spmm<LPR>(d_seg_row, d_seg_beg, d_seg_len, d_seg_slot, n_segs, d_long_row, d_long_first, \
                                               d_long_count, n_long, d_indices, d_values, d_X, d_Y, d_partial, d_addend,       \
                                               addend_scale, d_accum, d_accum_init, d_x_row_mask, d_y_row_mask, d_addend_row_mask, st)
    switch (ld) {
        case 32: QREC_SPMM(8);
        case 64: QREC_SPMM(16);
        case 128: QREC_SPMM(32);
        case 256: QREC_SPMM(64);
        default: set_error("qrec_spmm_csr: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_SPMM
}

int qrec_mark_batch_rows(const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int32_t B, int32_t n_users,
                         uint32_t *d_row_mask, void *stream) {
    QREC_REQUIRE(d_row_mask && B >= 0 && n_users >= 0, "qrec_mark_batch_rows: bad argument");
    QREC_REQUIRE(B == 0 || (d_u && d_i && d_j), "qrec_mark_batch_rows: null index array");
    if (B == 0) return QREC_OK;
    hipLaunchKernelGGL(mark_batch_rows_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, as_stream(stream), d_u, d_i,
                       d_j, B, n_users, d_row_mask);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_bpr_batch_loss_grad(const float *d_S, float div, int32_t n_users, int64_t n_rows, int32_t ld,
                             const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int32_t B, float eps,
                             float reg, float *d_dE, double *d_loss, uint32_t *d_row_mask, void *d_ordered_ws,
                             int64_t ordered_ws_bytes, void *stream) {
    QREC_REQUIRE(d_S && d_dE && d_loss && B >= 0 && div != 0.f, "qrec_bpr_batch_loss_grad: bad argument");
    QREC_REQUIRE(B == 0 || (d_u && d_i && d_j), "qrec_bpr_batch_loss_grad: null index array");
    QREC_REQUIRE(n_rows * (int64_t)ld * 4 < ((int64_t)1 << 32), "qrec_bpr_batch_loss_grad: table exceeds 4 GiB");
    if (B == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    const uint32_t bytes = (uint32_t)(n_rows * ld * 4);
    OrderedScatterWs ow = {};
    if (d_ordered_ws) {
        QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_bpr_batch_loss_grad: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
        const int rc = ordered_ws_carve(d_ordered_ws, ordered_ws_bytes, 3 * (int64_t)B, ld, &ow);
        if (rc != QREC_OK) return rc;
    }
#define QREC_BB(LPR, E)                                                                                          \
    blocks = (B + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 256) blocks = 256;                          \
    if (d_ordered_ws)                                                                                              \
        hipLaunchKernelGGL((bpr_batch_kernel<LPR, E, true>), dim3((unsigned)blocks), dim3(256), 0, st, d_S, div, n_users, d_u, d_i, d_j, B, \
                           eps, reg, d_dE, bytes, d_loss, d_row_mask, ow.contrib, ow.keys);                        \
    else                                                                                                           \
        hipLaunchKernelGGL((bpr_batch_kernel<LPR, E, false>), dim3((unsigned)blocks), dim3(256), 0, st, d_S, div, n_users, d_u, d_i, d_j, B, \
                           eps, reg, d_dE, bytes, d_loss, d_row_mask, (float *)nullptr, (int32_t *)nullptr)
    int64_t blocks;
    switch (ld) {
        case 32: QREC_BB(16, 2); break;
        case 64: QREC_BB(16, 4); break;
        case 128: QREC_BB(32, 4); break;
        case 256: QREC_BB(64, 4); break;
        default: set_error("qrec_bpr_batch_loss_grad: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_BB
    QREC_LAUNCH_CHECK();
    if (d_ordered_ws) return ordered_scatter_run(ow, 3 * (int64_t)B, ld, B, d_dE, st);     // one class per lookup (u, i, j)
    return QREC_OK;
}

int qrec_l2norm_rows_accum(const float *d_X, int64_t n_rows, int32_t ld, float *d_S, float *d_inv, void *stream) {
    QREC_REQUIRE(d_X && d_S && d_inv && n_rows >= 0, "qrec_l2norm_rows_accum: bad argument");
    if (n_rows == 0) return QREC_OK;
    int64_t blocks;
#define QREC_LN(LPR)                                                                                                  \
    blocks = (n_rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 4096) blocks = 4096;                       \
    hipLaunchKernelGGL((l2norm_accum_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_X, n_rows, d_S, d_inv)
    switch (ld) {
        case 32: QREC_LN(8); break;
        case 64: QREC_LN(16); break;
        case 128: QREC_LN(32); break;
        case 256: QREC_LN(64); break;
        default: set_error("qrec_l2norm_rows_accum: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_LN
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_l2norm_rows_bwd(const float *d_X, const float *d_inv, const float *d_dS, int64_t n_rows, int32_t ld, float *d_out,
                         void *stream) {
    QREC_REQUIRE(d_X && d_inv && d_dS && d_out && n_rows >= 0, "qrec_l2norm_rows_bwd: bad argument");
    if (n_rows == 0) return QREC_OK;
    int64_t blocks;
#define QREC_LB(LPR)                                                                                                  \
    blocks = (n_rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 4096) blocks = 4096;                       \
    hipLaunchKernelGGL((l2norm_bwd_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_X, d_inv, d_dS, n_rows, d_out)
    switch (ld) {
        case 32: QREC_LB(8); break;
        case 64: QREC_LB(16); break;
        case 128: QREC_LB(32); break;
        case 256: QREC_LB(64); break;
        default: set_error("qrec_l2norm_rows_bwd: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_LB
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_scale_copy(float *d_dst, const float *d_src, int64_t n_elems, float alpha, void *stream) {
    QREC_REQUIRE(d_dst && d_src && n_elems >= 0 && n_elems % 4 == 0, "qrec_scale_copy: bad argument (element count must be a multiple of 4)");
    if (n_elems == 0) return QREC_OK;
    int64_t blocks = (n_elems / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_dst, d_src, n_elems / 4, alpha);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_adam_step(float *d_theta, float *d_m, float *d_v, const float *d_grad, int64_t n_elems, float grad_scale,
                   float grad_l2, float alpha, float beta1, float beta2, float eps, void *stream) {
    QREC_REQUIRE(d_theta && d_m && d_v && d_grad && n_elems >= 0, "qrec_adam_step: bad argument");
    QREC_REQUIRE(n_elems % 4 == 0, "qrec_adam_step: element count must be a multiple of 4 (row stride is)");
    if (n_elems == 0) return QREC_OK;
    int64_t blocks = (n_elems / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_theta, d_m, d_v,
                       d_grad, n_elems / 4, grad_scale, grad_l2, alpha, beta1, beta2, eps);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
