// MHCN (model/ranking/MHCN.py:93-216) pieces that are not plain propagation: self-gating, channel attention and the
// hierarchical mutual-information loss, forward and backward.  All tables are [rows][ld] fp32, ld in {32, 64, 128, 256},
// columns >= d zero; one group of LPR = ld/4 lanes per row, float4 per lane.  The d x d products (2 n d^2 FLOP, a few
// hundred MFLOP per step) run on the vector ALUs with the weight matrix in LDS: they are bound by streaming the rows.
//
//   gate_fwd_kernel      Y = X * sigmoid(X W + b), S = sigmoid(.)                       MHCN.py:109-112
//   gate_bwd_kernel      Q = dY * X * S(1-S);  dX (+)= dY * S + Q W^T   (dW = X^T Q, db = colsum Q: qrec_buir_wgrad)
//   att_vec_kernel       v = M a^T   (sum(a * (e M), 1) = e . v)                        MHCN.py:113-121
//   att_fwd_kernel       score = softmax_k(e_k . v);  out = sum_k score_k e_k (+ half / 2)
//   att_bwd_kernel       de_k (+)= score_k dOut + dw_k v;  dv += sum dw_k e_k;  dhalf += dOut / 2
//   att_param_kernel     gM += dv (x) a;  ga += M^T dv
//   col_mean_kernel      graph = mean over rows                                         MHCN.py:202
//   hss_coef_kernel      per-row scores of the local / global MIM terms, their loss and d loss / d score; d graph
//   hss_grad_kernel      d em, d edge from those coefficients, through the row and column shuffles (inverse permutations)
//   random permutations  Philox keys + radix sort (rocPRIM) for the row shuffles, Fisher-Yates for the d columns
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

using namespace qrec;

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void philox10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

template <int LPR>
__device__ __forceinline__ float group_bcast(const f32x4 &v, int j, int lane) {
    // column j of the row held by this lane's group: lane (j >> 2) of the group, component j & 3
    const int src = (lane & ~(LPR - 1)) + (j >> 2);
    const int c = j & 3;
    const float mine = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
    return __shfl(mine, src, kWave);
}

// acc += row(x) . B, B = s_b[j][4r .. 4r+3] (LD x LD in LDS)
template <int LPR>
__device__ __forceinline__ f32x4 row_times_matrix(const f32x4 &x, const float *s_b, f32x4 acc, int lane, int r) {
    constexpr int LD = 4 * LPR;
#pragma unroll 4
    for (int j4 = 0; j4 < LPR; j4++) {
        const int src = (lane & ~(LPR - 1)) + j4;
        const float x0 = __shfl(x.x, src, kWave), x1 = __shfl(x.y, src, kWave), x2 = __shfl(x.z, src, kWave), x3 = __shfl(x.w, src, kWave);
        const float *b = s_b + (4 * j4) * LD + 4 * r;
        acc = acc + x0 * *reinterpret_cast<const f32x4 *>(b) + x1 * *reinterpret_cast<const f32x4 *>(b + LD) +
              x2 * *reinterpret_cast<const f32x4 *>(b + 2 * LD) + x3 * *reinterpret_cast<const f32x4 *>(b + 3 * LD);
    }
    return acc;
}

template <int LPR>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float *__restrict__ X, const float *__restrict__ W,
                                                       const float *__restrict__ bias, int64_t n, float *__restrict__ Y,
                                                       float *__restrict__ S) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    extern __shared__ float s_w[];
    for (int k = threadIdx.x; k < LD * LD / 4; k += blockDim.x) reinterpret_cast<f32x4 *>(s_w)[k] = reinterpret_cast<const f32x4 *>(W)[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + 4 * r);
    const int64_t n_iter = (n + GPW - 1) / GPW;         // whole wavefronts iterate together: the shuffles need every lane
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_iter; it += (int64_t)gridDim.x * 4) {
        const int64_t row = it * GPW + g;
        const bool live = row < n;
        const int64_t off = (live ? row : n - 1) * LD + 4 * r;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(X + off);
        const f32x4 z = row_times_matrix<LPR>(x, s_w, b4, lane, r);
        const f32x4 s = {sigmoidf_(z.x), sigmoidf_(z.y), sigmoidf_(z.z), sigmoidf_(z.w)};
        if (live) {
            *reinterpret_cast<f32x4 *>(S + off) = s;
            *reinterpret_cast<f32x4 *>(Y + off) = x * s;
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float *__restrict__ X, const float *__restrict__ S,
                                                       const float *__restrict__ dY, const float *__restrict__ W, int d, int64_t n,
                                                       float dy_scale, float *__restrict__ Q, float *__restrict__ dX, int accumulate) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    extern __shared__ float s_wt[];                     // W^T: s_wt[j][c] = W[c][j]
    for (int k = threadIdx.x; k < LD * LD; k += blockDim.x) s_wt[(k % LD) * LD + k / LD] = W[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t n_iter = (n + GPW - 1) / GPW;
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_iter; it += (int64_t)gridDim.x * 4) {
        const int64_t row = it * GPW + g;
        const bool live = row < n;
        const int64_t off = (live ? row : n - 1) * LD + 4 * r;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(X + off), s = *reinterpret_cast<const f32x4 *>(S + off);
        const f32x4 dy = dy_scale * *reinterpret_cast<const f32x4 *>(dY + off);
        const f32x4 one = {1.f, 1.f, 1.f, 1.f};
        f32x4 q = dy * x * s * (one - s);
        // the bias occupies the pad columns' sigmoid(b) otherwise: columns >= d carry no gradient
        if (4 * r + 0 >= d) q.x = 0.f;
        if (4 * r + 1 >= d) q.y = 0.f;
        if (4 * r + 2 >= d) q.z = 0.f;
        if (4 * r + 3 >= d) q.w = 0.f;
        f32x4 acc = dy * s;
        if (accumulate) acc = acc + *reinterpret_cast<const f32x4 *>(dX + off);
        acc = row_times_matrix<LPR>(q, s_wt, acc, lane, r);
        if (live) {
            *reinterpret_cast<f32x4 *>(Q + off) = q;
            *reinterpret_cast<f32x4 *>(dX + off) = acc;
        }
    }
}

// ---- column sums over rows without float atomics (round 6): every group's f32x4 partial through LDS, added in group order by
// the first LD threads -> part[block][256]; colsum_finish_kernel adds the blocks in block order.  The row -> group assignment is
// a function of (n, grid) alone, so the sums come out with the same bits on every launch (MHCN's attention / global-MIM
// gradients used LDS + global float atomics before: right to rounding, different from launch to launch).
constexpr int kColBlocks = 512;         // the launches below cap their grids here
template <int LPR>
__device__ __forceinline__ void block_colsum_store(const f32x4 &mine, float *s_part, float *__restrict__ part) {
    constexpr int LD = 4 * LPR, NG = 256 / LPR;
    *reinterpret_cast<f32x4 *>(s_part + 4 * threadIdx.x) = mine;          // thread = group * LPR + r holds columns 4r .. 4r+3
    __syncthreads();
    if (threadIdx.x < LD) {
        float t = 0.f;
#pragma unroll 4
        for (int g = 0; g < NG; g++) t += s_part[(g * LPR + (threadIdx.x >> 2)) * 4 + (threadIdx.x & 3)];
        part[(int64_t)blockIdx.x * 256 + threadIdx.x] = t;
    }
}
// out[c] (+)= scale * sum_b part[b][c], blocks in order
__global__ void colsum_finish_kernel(const float *__restrict__ part, int n_blocks, int ld, float scale, int accumulate, float *__restrict__ out) {
    const int c = threadIdx.x;
    if (c >= ld) return;
    float t = 0.f;
    for (int b = 0; b < n_blocks; b++) t += part[(int64_t)b * 256 + c];
    out[c] = accumulate ? out[c] + scale * t : scale * t;
}

__global__ void att_vec_kernel(const float *__restrict__ M, const float *__restrict__ a, int ld, float *__restrict__ v) {
    const int j = threadIdx.x;
    if (j >= ld) return;
    float s = 0.f;
    for (int c = 0; c < ld; c++) s += M[j * ld + c] * a[c];
    v[j] = s;
}

template <int LPR>
__global__ __launch_bounds__(256) void att_fwd_kernel(const float *__restrict__ e1, const float *__restrict__ e2,
                                                      const float *__restrict__ e3, const float *__restrict__ v,
                                                      const float *__restrict__ half, int64_t n, float *__restrict__ score,
                                                      float *__restrict__ out) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const f32x4 v4 = *reinterpret_cast<const f32x4 *>(v + 4 * r);
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n; row += (int64_t)gridDim.x * 4 * GPW) {
        const int64_t off = row * LD + 4 * r;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(e1 + off), b = *reinterpret_cast<const f32x4 *>(e2 + off),
                    c = *reinterpret_cast<const f32x4 *>(e3 + off);
        float wa = a.x * v4.x + a.y * v4.y + a.z * v4.z + a.w * v4.w, wb = b.x * v4.x + b.y * v4.y + b.z * v4.z + b.w * v4.w,
              wc = c.x * v4.x + c.y * v4.y + c.z * v4.z + c.w * v4.w;
        wa = row_allreduce_sum<LPR>(wa); wb = row_allreduce_sum<LPR>(wb); wc = row_allreduce_sum<LPR>(wc);
        const float m = fmaxf(wa, fmaxf(wb, wc));
        const float ea = expf(wa - m), eb = expf(wb - m), ec = expf(wc - m), tot = ea + eb + ec;
        const float sa = ea / tot, sb = eb / tot, sc = ec / tot;
        f32x4 o = sa * a + sb * b + sc * c;
        if (half) o = o + *reinterpret_cast<const f32x4 *>(half + off) / 2.0f;
        *reinterpret_cast<f32x4 *>(out + off) = o;
        if (r == 0) { const f32x4 s4 = {sa, sb, sc, 0.f}; *reinterpret_cast<f32x4 *>(score + row * 4) = s4; }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void att_bwd_kernel(const float *__restrict__ dOut, const float *__restrict__ e1,
                                                      const float *__restrict__ e2, const float *__restrict__ e3,
                                                      const float *__restrict__ score, const float *__restrict__ v, int64_t n,
                                                      float *__restrict__ de1, float *__restrict__ de2, float *__restrict__ de3,
                                                      int accumulate, float *__restrict__ dhalf, int half_accumulate,
                                                      float *__restrict__ dv_part) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    __shared__ float s_dv[1024];
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const f32x4 v4 = *reinterpret_cast<const f32x4 *>(v + 4 * r);
    f32x4 dvl = {0.f, 0.f, 0.f, 0.f};
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n; row += (int64_t)gridDim.x * 4 * GPW) {
        const int64_t off = row * LD + 4 * r;
        const f32x4 dO = *reinterpret_cast<const f32x4 *>(dOut + off);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(e1 + off), b = *reinterpret_cast<const f32x4 *>(e2 + off),
                    c = *reinterpret_cast<const f32x4 *>(e3 + off);
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(score + row * 4);
        float da = dO.x * a.x + dO.y * a.y + dO.z * a.z + dO.w * a.w, db = dO.x * b.x + dO.y * b.y + dO.z * b.z + dO.w * b.w,
              dc = dO.x * c.x + dO.y * c.y + dO.z * c.z + dO.w * c.w;
        da = row_allreduce_sum<LPR>(da); db = row_allreduce_sum<LPR>(db); dc = row_allreduce_sum<LPR>(dc);
        const float mean = s4.x * da + s4.y * db + s4.z * dc;
        const float wa = s4.x * (da - mean), wb = s4.y * (db - mean), wc = s4.z * (dc - mean);
        f32x4 ga = s4.x * dO + wa * v4, gb = s4.y * dO + wb * v4, gc = s4.z * dO + wc * v4;
        if (accumulate) {
            ga = ga + *reinterpret_cast<const f32x4 *>(de1 + off); gb = gb + *reinterpret_cast<const f32x4 *>(de2 + off);
            gc = gc + *reinterpret_cast<const f32x4 *>(de3 + off);
        }
        *reinterpret_cast<f32x4 *>(de1 + off) = ga; *reinterpret_cast<f32x4 *>(de2 + off) = gb; *reinterpret_cast<f32x4 *>(de3 + off) = gc;
        if (dhalf) {
            f32x4 h = dO / 2.0f;
            if (half_accumulate) h = h + *reinterpret_cast<const f32x4 *>(dhalf + off);
            *reinterpret_cast<f32x4 *>(dhalf + off) = h;
        }
        dvl = dvl + wa * a + wb * b + wc * c;
    }
    block_colsum_store<LPR>(dvl, s_dv, dv_part);
}

// gM[j][c] += dv[j] a[c];  ga[c] += sum_j dv[j] M[j][c]      (one block of ld x ... threads; tiny)
__global__ void att_param_kernel(const float *__restrict__ dv, const float *__restrict__ M, const float *__restrict__ a, int ld,
                                 float *__restrict__ gM, float *__restrict__ ga) {
    for (int k = threadIdx.x; k < ld * ld; k += blockDim.x) gM[k] += dv[k / ld] * a[k % ld];
    for (int c = threadIdx.x; c < ld; c += blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < ld; j++) s += dv[j] * M[j * ld + c];
        ga[c] += s;
    }
}

// part[block][c] = sum of this block's rows of X[.][c]   (colsum_finish_kernel adds the blocks and scales)
template <int LPR>
__global__ __launch_bounds__(256) void col_sum_kernel(const float *__restrict__ X, int64_t n, float *__restrict__ part) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    __shared__ float s_acc[1024];
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n; row += (int64_t)gridDim.x * 4 * GPW)
        acc = acc + *reinterpret_cast<const f32x4 *>(X + row * LD + 4 * r);
    block_colsum_store<LPR>(acc, s_acc, part);
}

// hierarchical_self_supervision (MHCN.py:184-206) scores per row r:
//   pos = em[r].edge[r]; neg1 = em[p1[r]].edge[r]; neg2 = edge[p2[r]][k2[.]].em[r]
//   pg = edge[r].graph;  ng = edge[p3[r]][k3[.]].graph
//   loss += -log sig(pos-neg1) - log sig(neg1-neg2) - log sig(pg-ng)
//   coef[r] = {d/dpos, d/dneg1, d/dneg2, c3 = d/dpg = -d/dng};  dgraph += c3 (edge[r] - edge[p3[r]][k3[.]])
template <int LPR>
__global__ __launch_bounds__(256) void hss_coef_kernel(const float *__restrict__ em, const float *__restrict__ edge,
                                                       const int32_t *__restrict__ p1, const int32_t *__restrict__ p2,
                                                       const int32_t *__restrict__ k2, const int32_t *__restrict__ p3,
                                                       const int32_t *__restrict__ k3, const float *__restrict__ graph, int d,
                                                       int64_t n, float *__restrict__ coef, float *__restrict__ dgraph_part,
                                                       double *__restrict__ loss_out) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    __shared__ float s_dg[1024];
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const f32x4 gr = *reinterpret_cast<const f32x4 *>(graph + 4 * r);
    int kc2[4], kc3[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { const int col = 4 * r + c; kc2[c] = col < d ? k2[col] : col; kc3[c] = col < d ? k3[col] : col; }
    f32x4 dgl = {0.f, 0.f, 0.f, 0.f};
    double loss = 0.0;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n; row += (int64_t)gridDim.x * 4 * GPW) {
        const int64_t off = row * LD + 4 * r;
        const f32x4 m = *reinterpret_cast<const f32x4 *>(em + off), e = *reinterpret_cast<const f32x4 *>(edge + off);
        const f32x4 m1 = *reinterpret_cast<const f32x4 *>(em + (int64_t)p1[row] * LD + 4 * r);
        const float *r2 = edge + (int64_t)p2[row] * LD, *r3 = edge + (int64_t)p3[row] * LD;
        const f32x4 e2 = {r2[kc2[0]], r2[kc2[1]], r2[kc2[2]], r2[kc2[3]]}, e3 = {r3[kc3[0]], r3[kc3[1]], r3[kc3[2]], r3[kc3[3]]};
        auto dot = [](const f32x4 &a, const f32x4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
        float pos = dot(m, e), neg1 = dot(m1, e), neg2 = dot(e2, m), pg = dot(e, gr), ng = dot(e3, gr);
        pos = row_allreduce_sum<LPR>(pos); neg1 = row_allreduce_sum<LPR>(neg1); neg2 = row_allreduce_sum<LPR>(neg2);
        pg = row_allreduce_sum<LPR>(pg); ng = row_allreduce_sum<LPR>(ng);
        const float s1 = sigmoidf_(pos - neg1), s2 = sigmoidf_(neg1 - neg2), s3 = sigmoidf_(pg - ng);
        const float c1 = -(1.f - s1), c2 = -(1.f - s2), c3 = -(1.f - s3);
        if (r == 0) {
            const f32x4 cf = {c1, c2 - c1, -c2, c3};
            *reinterpret_cast<f32x4 *>(coef + row * 4) = cf;
            loss += (double)(-logf(s1)) + (double)(-logf(s2)) + (double)(-logf(s3));
        }
        dgl = dgl + c3 * (e - e3);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) loss += __shfl_xor(loss, m, kWave);
    if (lane == 0 && loss != 0.0) atomicAdd(loss_out, loss);
    block_colsum_store<LPR>(dgl, s_dg, dgraph_part);
}

// Gradients of the block above w.r.t. em and edge, gathered through the inverse permutations (q = p^-1):
//   dem[r]      = scale * ( cpos[r] edge[r] + cneg2[r] edge[p2[r]][k2[.]] + cneg1[q1[r]] edge[q1[r]] )
//   dedge[r][c] = scale * ( cpos[r] em[r][c] + cneg1[r] em[p1[r]][c] + c3[r] graph[c] + cneg2[q2[r]] em[q2[r]][k2inv[c]]
//                           - c3[q3[r]] graph[k3inv[c]] + dgraph[c] / n )
template <int LPR>
__global__ __launch_bounds__(256) void hss_grad_kernel(const float *__restrict__ em, const float *__restrict__ edge,
                                                       const float *__restrict__ coef, const int32_t *__restrict__ p1,
                                                       const int32_t *__restrict__ q1, const int32_t *__restrict__ p2,
                                                       const int32_t *__restrict__ q2, const int32_t *__restrict__ k2,
                                                       const int32_t *__restrict__ k2inv, const int32_t *__restrict__ q3,
                                                       const int32_t *__restrict__ k3inv, const float *__restrict__ graph,
                                                       const float *__restrict__ dgraph, int d, int64_t n, float scale,
                                                       float *__restrict__ dem, float *__restrict__ dedge) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const f32x4 gr = *reinterpret_cast<const f32x4 *>(graph + 4 * r);
    const f32x4 dg = *reinterpret_cast<const f32x4 *>(dgraph + 4 * r) / (float)n;
    int kc2[4], ki2[4], ki3[4];
    bool in_d[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int col = 4 * r + c;
        in_d[c] = col < d;
        kc2[c] = in_d[c] ? k2[col] : col; ki2[c] = in_d[c] ? k2inv[col] : col; ki3[c] = in_d[c] ? k3inv[col] : col;
    }
    const f32x4 g3 = {graph[ki3[0]], graph[ki3[1]], graph[ki3[2]], graph[ki3[3]]};
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; row < n; row += (int64_t)gridDim.x * 4 * GPW) {
        const int64_t off = row * LD + 4 * r;
        const f32x4 cf = *reinterpret_cast<const f32x4 *>(coef + row * 4);             // {pos, neg1, neg2, c3}
        const int a1 = q1[row], a2 = q2[row], a3 = q3[row];
        const f32x4 m = *reinterpret_cast<const f32x4 *>(em + off), e = *reinterpret_cast<const f32x4 *>(edge + off);
        const float *r2 = edge + (int64_t)p2[row] * LD;
        const f32x4 e2 = {r2[kc2[0]], r2[kc2[1]], r2[kc2[2]], r2[kc2[3]]};
        const f32x4 eq1 = *reinterpret_cast<const f32x4 *>(edge + (int64_t)a1 * LD + 4 * r);
        const f32x4 m1 = *reinterpret_cast<const f32x4 *>(em + (int64_t)p1[row] * LD + 4 * r);
        const float *mq2 = em + (int64_t)a2 * LD;
        const f32x4 m2 = {mq2[ki2[0]], mq2[ki2[1]], mq2[ki2[2]], mq2[ki2[3]]};
        const float cn1_q1 = coef[(int64_t)a1 * 4 + 1], cn2_q2 = coef[(int64_t)a2 * 4 + 2], c3_q3 = coef[(int64_t)a3 * 4 + 3];
        f32x4 de = cf.x * e + cf.z * e2 + cn1_q1 * eq1;
        f32x4 dd = cf.x * m + cf.y * m1 + cf.w * gr + cn2_q2 * m2 - c3_q3 * g3 + dg;
        if (!in_d[0]) { de.x = 0.f; dd.x = 0.f; }
        if (!in_d[1]) { de.y = 0.f; dd.y = 0.f; }
        if (!in_d[2]) { de.z = 0.f; dd.z = 0.f; }
        if (!in_d[3]) { de.w = 0.f; dd.w = 0.f; }
        *reinterpret_cast<f32x4 *>(dem + off) = scale * de;
        *reinterpret_cast<f32x4 *>(dedge + off) = scale * dd;
    }
}

// ---- random permutations (tf.random.shuffle of range(n)) --------------------------------------------------------------
// `count` permutations of range(n) from ONE sort: key = (permutation number << 40) | 40 random bits, value = position
// inside the permutation; after the sort, slots [s n, (s+1) n) hold permutation s (ties -- 2^-40 per pair -- keep index order)
__global__ void perm_keys_kernel(int64_t n, int64_t total, uint64_t seed, uint64_t stream_id, uint64_t *__restrict__ keys,
                                 int32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    philox10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t seg = (uint64_t)(i / n);
    keys[i] = (seg << 40) | ((((uint64_t)c[0] << 32) | c[1]) >> 24);
    idx[i] = (int32_t)(i % n);
}
__global__ void invert_perm_kernel(const int32_t *__restrict__ p, int64_t n, int64_t total, int32_t *__restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) inv[(i / n) * n + p[i]] = (int32_t)(i % n);
}
// Fisher-Yates on a few hundred elements: one thread per permutation (count of them), 24-bit Philox draws
__global__ void small_perm_kernel(int32_t n, int32_t count, uint64_t seed, uint64_t stream_id, int32_t *__restrict__ perms,
                                  int32_t *__restrict__ invs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    int32_t *p = perms + (int64_t)t * n, *q = invs + (int64_t)t * n;
    for (int i = 0; i < n; i++) p[i] = i;
    for (int i = n - 1; i >= 1; i--) {
        uint32_t c[4] = {(uint32_t)i, (uint32_t)t, (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
        philox10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u);
        const int j = (int)(((uint64_t)c[0] * (uint64_t)(i + 1)) >> 32);
        const int32_t tmp = p[i]; p[i] = p[j]; p[j] = tmp;
    }
    for (int i = 0; i < n; i++) q[p[i]] = i;
}

template <int LPR>
int launch_rows(int64_t n) {
    int64_t blocks = (n + 4 * (64 / LPR) - 1) / (4 * (64 / LPR));
    return (int)(blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks));
}

}  // namespace

#define QREC_MHCN_LD_SWITCH(NAME, CALL)                                                                                  \
    switch (ld) {                                                                                                        \
        case 32: { constexpr int LPR = 8; CALL; } break;                                                                 \
        case 64: { constexpr int LPR = 16; CALL; } break;                                                                \
        case 128: { constexpr int LPR = 32; CALL; } break;                                                               \
        case 256: { constexpr int LPR = 64; CALL; } break;                                                               \
        default: set_error(NAME ": row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID; \
    }

extern "C" {

int qrec_gate_fwd(const float *d_X, const float *d_W, const float *d_bias, int64_t n_rows, int32_t ld, float *d_Y, float *d_S,
                  void *stream) {
    QREC_REQUIRE(d_X && d_W && d_bias && d_Y && d_S && n_rows >= 0, "qrec_gate_fwd: bad argument");
    if (n_rows == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)ld * ld * sizeof(float);
    QREC_MHCN_LD_SWITCH("qrec_gate_fwd", {
        if (lds > 64 * 1024) QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gate_fwd_kernel<LPR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int blocks = launch_rows<LPR>(n_rows); if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL((gate_fwd_kernel<LPR>), dim3((unsigned)blocks), dim3(256), lds, st, d_X, d_W, d_bias, n_rows, d_Y, d_S);
    })
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_gate_bwd(const float *d_X, const float *d_S, const float *d_dY, const float *d_W, int64_t n_rows, int32_t d, int32_t ld,
                  float dy_scale, float *d_Q, float *d_dX, int32_t accumulate, void *stream) {
    QREC_REQUIRE(d_X && d_S && d_dY && d_W && d_Q && d_dX && n_rows >= 0 && d >= 1 && d <= ld, "qrec_gate_bwd: bad argument");
    if (n_rows == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)ld * ld * sizeof(float);
    QREC_MHCN_LD_SWITCH("qrec_gate_bwd", {
        if (lds > 64 * 1024) QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gate_bwd_kernel<LPR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int blocks = launch_rows<LPR>(n_rows); if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL((gate_bwd_kernel<LPR>), dim3((unsigned)blocks), dim3(256), lds, st, d_X, d_S, d_dY, d_W, d, n_rows, dy_scale, d_Q, d_dX, accumulate);
    })
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_channel_attention_fwd(const float *d_e1, const float *d_e2, const float *d_e3, const float *d_att, const float *d_att_mat,
                               const float *d_half, int64_t n_rows, int32_t ld, float *d_v, float *d_score, float *d_out,
                               void *stream) {
    QREC_REQUIRE(d_e1 && d_e2 && d_e3 && d_att && d_att_mat && d_v && d_score && d_out && n_rows >= 0, "qrec_channel_attention_fwd: bad argument");
    hipStream_t st = as_stream(stream);
    QREC_REQUIRE(ld >= 1 && ld <= 256, "qrec_channel_attention_fwd: bad row stride");
    hipLaunchKernelGGL(att_vec_kernel, dim3(1), dim3(256), 0, st, d_att_mat, d_att, ld, d_v);
    QREC_LAUNCH_CHECK();
    if (n_rows == 0) return QREC_OK;
    QREC_MHCN_LD_SWITCH("qrec_channel_attention_fwd", {
        hipLaunchKernelGGL((att_fwd_kernel<LPR>), dim3((unsigned)launch_rows<LPR>(n_rows)), dim3(256), 0, st, d_e1, d_e2, d_e3, d_v, d_half, n_rows, d_score, d_out);
    })
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_channel_attention_bwd(const float *d_dOut, const float *d_e1, const float *d_e2, const float *d_e3, const float *d_score,
                               const float *d_v, const float *d_att, const float *d_att_mat, int64_t n_rows, int32_t ld,
                               float *d_de1, float *d_de2, float *d_de3, int32_t accumulate, float *d_dhalf, int32_t half_accumulate,
                               float *d_dv_scratch, float *d_g_att, float *d_g_att_mat, void *stream) {
    QREC_REQUIRE(d_dOut && d_e1 && d_e2 && d_e3 && d_score && d_v && d_att && d_att_mat && d_de1 && d_de2 && d_de3 && d_dv_scratch &&
                 d_g_att && d_g_att_mat && n_rows >= 0, "qrec_channel_attention_bwd: bad argument");
    hipStream_t st = as_stream(stream);
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_channel_attention_bwd: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    int blocks = 0;
    if (n_rows > 0) {
        QREC_MHCN_LD_SWITCH("qrec_channel_attention_bwd", {
            blocks = launch_rows<LPR>(n_rows); if (blocks > kColBlocks) blocks = kColBlocks;
            hipLaunchKernelGGL((att_bwd_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, d_dOut, d_e1, d_e2, d_e3, d_score, d_v, n_rows,
                               d_de1, d_de2, d_de3, accumulate, d_dhalf, half_accumulate, d_dv_scratch + 256);
        })
        QREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(1), dim3(256), 0, st, d_dv_scratch + 256, blocks, ld, 1.0f, 0, d_dv_scratch);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(att_param_kernel, dim3(1), dim3(256), 0, st, d_dv_scratch, d_att_mat, d_att, ld, d_g_att_mat, d_g_att);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_hss_loss_grad(const float *d_em, const float *d_edge, int64_t n_rows, int32_t d, int32_t ld, const int32_t *d_p1,
                       const int32_t *d_p1inv, const int32_t *d_p2, const int32_t *d_p2inv, const int32_t *d_k2,
                       const int32_t *d_k2inv, const int32_t *d_p3, const int32_t *d_p3inv, const int32_t *d_k3,
                       const int32_t *d_k3inv, float scale, float *d_scratch, float *d_dem, float *d_dedge, double *d_loss,
                       void *stream) {
    QREC_REQUIRE(d_em && d_edge && d_p1 && d_p1inv && d_p2 && d_p2inv && d_k2 && d_k2inv && d_p3 && d_p3inv && d_k3 && d_k3inv &&
                 d_scratch && d_dem && d_dedge && d_loss && n_rows >= 0 && d >= 1 && d <= ld, "qrec_hss_loss_grad: bad argument");
    if (n_rows == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    float *graph = d_scratch, *dgraph = d_scratch + 256, *part = d_scratch + 512, *coef = part + kColBlocks * 256;   // [256] [256] [512][256] [n][4]
    QREC_HIP_CHECK(hipMemsetAsync(d_scratch, 0, sizeof(float) * 512, st));       // the pad columns of graph / dgraph stay zero
    QREC_MHCN_LD_SWITCH("qrec_hss_loss_grad", {
        int blocks = launch_rows<LPR>(n_rows); if (blocks > kColBlocks) blocks = kColBlocks;
        hipLaunchKernelGGL((col_sum_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, d_edge, n_rows, part);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3(1), dim3(256), 0, st, part, blocks, ld, 1.0f / (float)n_rows, 0, graph);
        hipLaunchKernelGGL((hss_coef_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, d_em, d_edge, d_p1, d_p2, d_k2, d_p3, d_k3, graph, d,
                           n_rows, coef, part, d_loss);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3(1), dim3(256), 0, st, part, blocks, ld, 1.0f, 0, dgraph);
        hipLaunchKernelGGL((hss_grad_kernel<LPR>), dim3((unsigned)launch_rows<LPR>(n_rows)), dim3(256), 0, st, d_em, d_edge, coef, d_p1, d_p1inv,
                           d_p2, d_p2inv, d_k2, d_k2inv, d_p3inv, d_k3inv, graph, dgraph, d, n_rows, scale, d_dem, d_dedge);
    })
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_hss_scratch_bytes(int64_t n_rows, int64_t *bytes) {
    QREC_REQUIRE(bytes && n_rows >= 0, "qrec_hss_scratch_bytes: bad argument");
    *bytes = (int64_t)sizeof(float) * (512 + kColBlocks * 256 + 4 * n_rows);
    return QREC_OK;
}

int qrec_channel_attention_scratch_bytes(int64_t *bytes) {
    QREC_REQUIRE(bytes, "qrec_channel_attention_scratch_bytes: bad argument");
    *bytes = (int64_t)sizeof(float) * (256 + kColBlocks * 256);       // dv, then the blocks' partial column sums
    return QREC_OK;
}

int qrec_random_permutations_scratch_bytes(int64_t n, int32_t count, int64_t *bytes) {
    QREC_REQUIRE(bytes && n >= 0 && count >= 0 && count < (1 << 20) && n * count < ((int64_t)1 << 31), "qrec_random_permutations_scratch_bytes: bad argument");
    const size_t total = (size_t)n * (size_t)count;
    size_t tmp = 0;
    QREC_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, total));
    *bytes = (int64_t)(((tmp + 255) / 256) * 256 + total * (8 + 8 + 4));
    return QREC_OK;
}

int qrec_random_permutations(int64_t n, int32_t count, uint64_t seed, uint64_t stream_id, void *d_scratch, int32_t *d_perms,
                             int32_t *d_invs, void *stream) {
    QREC_REQUIRE(d_scratch && d_perms && n >= 0 && count >= 0 && count < (1 << 20) && n * count < ((int64_t)1 << 31), "qrec_random_permutations: bad argument");
    const int64_t total = n * count;
    if (total == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    size_t tmp = 0;
    QREC_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)total));
    char *base = static_cast<char *>(d_scratch);
    const size_t tmp_pad = ((tmp + 255) / 256) * 256;
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(base + tmp_pad), *keys_out = keys_in + total;
    int32_t *idx = reinterpret_cast<int32_t *>(keys_out + total);
    hipLaunchKernelGGL(perm_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n, total, seed, stream_id, keys_in, idx);
    QREC_LAUNCH_CHECK();
    QREC_HIP_CHECK(rocprim::radix_sort_pairs(base, tmp, keys_in, keys_out, idx, d_perms, (size_t)total, 0, 64, st));
    if (d_invs) {
        hipLaunchKernelGGL(invert_perm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_perms, n, total, d_invs);
        QREC_LAUNCH_CHECK();
    }
    return QREC_OK;
}

int qrec_small_permutations(int32_t n, int32_t count, uint64_t seed, uint64_t stream_id, int32_t *d_perms, int32_t *d_invs,
                            void *stream) {
    QREC_REQUIRE(d_perms && d_invs && n >= 1 && n <= 4096 && count >= 0, "qrec_small_permutations: bad argument");
    if (count == 0) return QREC_OK;
    hipLaunchKernelGGL(small_perm_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, as_stream(stream), n, count, seed, stream_id, d_perms, d_invs);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
