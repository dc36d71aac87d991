// BUIR (model/ranking/BUIR.py:13-172): what the model adds to the LightGCN propagation (graph.hip).
//
//   buir_batch_kernel   per batch element: q = tanh(x W + b) for the online rows x of user u and item i
//                       (BUIR.py:105-115), the two cosine terms against the TARGET rows (:127-129), and their
//                       gradients back to the online rows (scatter-added into dS) and to pre = xW + b
//   buir_wgrad_kernel   dW = X^T dPre, db = column sums, over the 2B (x, dpre) pairs the batch kernel stored --
//                       deterministic (fixed row partition, fixed combine order)
//   ema_kernel          target = target*tau + online*(1 - tau)                       (BUIR.py:120-123,159)
//
// Only batch rows of q are ever needed while training (embedding_lookup, BUIR.py:115-118), so the d x d linear layer
// is applied to <= 2B rows per step instead of to all N: a few MFLOP, done with the weights in LDS -- no MFMA.
#include "common.h"

using namespace qrec;

namespace {

// One group of LPR lanes per batch element; lane r owns columns [4r, 4r+4) (ld = 4*LPR).  LDS: W and W^T.
template <int LPR>
__global__ __launch_bounds__(256) void buir_batch_kernel(
    const float *__restrict__ S_on, const float *__restrict__ S_tar, float div, int n_users,
    const float *__restrict__ W, const float *__restrict__ bias, const int32_t *__restrict__ u_idx,
    const int32_t *__restrict__ i_idx, int B, float *__restrict__ dS, float *__restrict__ Xb,
    float *__restrict__ Gb, double *__restrict__ loss_out, float *__restrict__ contrib, int32_t *__restrict__ keys) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    extern __shared__ float s_w[];                       // [LD][LD] W, then [LD][LD+1] W^T (padded)
    float *s_wt = s_w + LD * LD;
    for (int k = threadIdx.x; k < LD * LD; k += blockDim.x) {
        const float w = W[k];
        s_w[k] = w;
        s_wt[(k % LD) * (LD + 1) + k / LD] = w;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    const int64_t n_groups = (int64_t)gridDim.x * 4 * GPW;
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + 4 * r);
    double loss = 0.0;
    for (int64_t b = gid; b < B; b += n_groups) {
        const int64_t ru = u_idx[b], ri = (int64_t)n_users + i_idx[b];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            // side 0: online row of the user against the target row of the item; side 1: the other way round
            const int64_t rx = side == 0 ? ru : ri, rt = side == 0 ? ri : ru;
            f32x4 x = *reinterpret_cast<const f32x4 *>(S_on + rx * LD + 4 * r);
            f32x4 t = *reinterpret_cast<const f32x4 *>(S_tar + rt * LD + 4 * r);
            x.x /= div; x.y /= div; x.z /= div; x.w /= div; t.x /= div; t.y /= div; t.z /= div; t.w /= div;
            // pre = x W + b : every lane needs all of x -> broadcast inside the group
            f32x4 pre = b4;
#pragma unroll 4
            for (int k4 = 0; k4 < LPR; k4++) {
                const float x0 = __shfl(x.x, g * LPR + k4, kWave), x1 = __shfl(x.y, g * LPR + k4, kWave);
                const float x2 = __shfl(x.z, g * LPR + k4, kWave), x3 = __shfl(x.w, g * LPR + k4, kWave);
                const float *wr = s_w + (4 * k4) * LD + 4 * r;
                pre = pre + x0 * *reinterpret_cast<const f32x4 *>(wr) + x1 * *reinterpret_cast<const f32x4 *>(wr + LD) +
                      x2 * *reinterpret_cast<const f32x4 *>(wr + 2 * LD) + x3 * *reinterpret_cast<const f32x4 *>(wr + 3 * LD);
            }
            f32x4 q;
            q.x = tanhf(pre.x); q.y = tanhf(pre.y); q.z = tanhf(pre.z); q.w = tanhf(pre.w);
            float sq = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, st = t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
            sq = row_allreduce_sum<LPR>(sq); st = row_allreduce_sum<LPR>(st);
            const float iq = 1.0f / sqrtf(fmaxf(sq, 1e-12f)), it = 1.0f / sqrtf(fmaxf(st, 1e-12f));   // tf.math.l2_normalize
            const f32x4 qh = q * iq, th = t * it;
            float c = qh.x * th.x + qh.y * th.y + qh.z * th.z + qh.w * th.w;
            c = row_allreduce_sum<LPR>(c);
            if (r == 0) loss += (double)((1.0f - c) * 0.5f);
            // d loss / d q = -(that - c qhat) / |q| / 2 ; through tanh: dpre = dq (1 - q^2)
            f32x4 dq = (th - qh * c) * (-0.5f * iq);
            f32x4 one_m;
            one_m.x = 1.f - q.x * q.x; one_m.y = 1.f - q.y * q.y; one_m.z = 1.f - q.z * q.z; one_m.w = 1.f - q.w * q.w;
            const f32x4 dp = dq * one_m;
            const int64_t slot = side == 0 ? b : (int64_t)B + b;
            *reinterpret_cast<f32x4 *>(Xb + slot * LD + 4 * r) = x;
            *reinterpret_cast<f32x4 *>(Gb + slot * LD + 4 * r) = dp;
            // dx = dpre W^T  (dx[c] = sum_k dpre[k] W[c][k] = sum_k dpre[k] WT[k][c])
            f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int k4 = 0; k4 < LPR; k4++) {
                const float d0 = __shfl(dp.x, g * LPR + k4, kWave), d1 = __shfl(dp.y, g * LPR + k4, kWave);
                const float d2 = __shfl(dp.z, g * LPR + k4, kWave), d3 = __shfl(dp.w, g * LPR + k4, kWave);
                const float *w0 = s_wt + (4 * k4) * (LD + 1) + 4 * r;
                dx.x += d0 * w0[0] + d1 * w0[LD + 1] + d2 * w0[2 * (LD + 1)] + d3 * w0[3 * (LD + 1)];
                dx.y += d0 * w0[1] + d1 * w0[LD + 2] + d2 * w0[2 * (LD + 1) + 1] + d3 * w0[3 * (LD + 1) + 1];
                dx.z += d0 * w0[2] + d1 * w0[LD + 3] + d2 * w0[2 * (LD + 1) + 2] + d3 * w0[3 * (LD + 1) + 2];
                dx.w += d0 * w0[3] + d1 * w0[LD + 4] + d2 * w0[2 * (LD + 1) + 3] + d3 * w0[3 * (LD + 1) + 3];
            }
            // d online-mean row (the 1/(L+1) of the mean is applied by the caller's Adam grad_scale): scatter-add
            if (contrib) {          // parity mode: slot's row into the ordered-scatter workspace (ordered.hip), added per row in slot order
                *reinterpret_cast<f32x4 *>(contrib + slot * LD + 4 * r) = dx;
                if (r == 0) keys[slot] = (int32_t)rx;
            } else {
                float *dst = dS + rx * LD + 4 * r;
                atomicAdd(dst + 0, dx.x); atomicAdd(dst + 1, dx.y); atomicAdd(dst + 2, dx.z); atomicAdd(dst + 3, dx.w);
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) loss += __shfl_xor(loss, m, kWave);
    if (lane == 0 && loss != 0.0) atomicAdd(loss_out, loss);
}

// gW[k][c] = sum_rows Xb[row][k] * Gb[row][c] (k < ld), gb[c] = sum_rows Gb[row][c] (k == ld).
// Stage 1: block (k, slice) adds its slice of the rows (kWgradSlices slices; 256 threads = (256/ld) interleaved row
// partitions x ld columns, combined in order through LDS) into part[slice][k][c].  Stage 2 adds the slices in order.
// (One block per k over all 4,000 rows was a 1,000-deep dependent-load loop: 298 us.)
constexpr int kWgradSlices = 16;
__global__ __launch_bounds__(256) void buir_wgrad_kernel(const float *__restrict__ Xb, const float *__restrict__ Gb,
                                                         int n_rows, int ld, float *__restrict__ part) {
    __shared__ float s_part[256];
    const int k = blockIdx.x, slice = blockIdx.y, c = threadIdx.x % ld, p = threadIdx.x / ld, n_parts = 256 / ld;
    const int per = (n_rows + kWgradSlices - 1) / kWgradSlices;
    const int r0 = slice * per, r1 = (r0 + per) < n_rows ? (r0 + per) : n_rows;
    float acc = 0.f;
#pragma unroll 8
    for (int row = r0 + p; row < r1; row += n_parts) {
        const float xv = k < ld ? Xb[(int64_t)row * ld + k] : 1.f;
        acc += xv * Gb[(int64_t)row * ld + c];
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (p == 0) {
        float tot = 0.f;
        for (int q = 0; q < n_parts; q++) tot += s_part[q * ld + c];
        part[((int64_t)slice * (ld + 1) + k) * ld + c] = tot;
    }
}
__global__ __launch_bounds__(256) void buir_wgrad_reduce_kernel(const float *__restrict__ part, int ld, float *__restrict__ gW,
                                                                float *__restrict__ gb) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (ld + 1) * ld) return;
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < kWgradSlices; s++) tot += part[(int64_t)s * (ld + 1) * ld + idx];
    if (idx < ld * ld) gW[idx] = tot; else gb[idx - ld * ld] = tot;
}

__global__ __launch_bounds__(256) void ema_kernel(float *__restrict__ target, const float *__restrict__ online, float tau,
                                                  int64_t n4) {
    const float om = 1.0f - tau;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 t = reinterpret_cast<const f32x4 *>(target)[k], o = reinterpret_cast<const f32x4 *>(online)[k];
        reinterpret_cast<f32x4 *>(target)[k] = t * tau + o * om;
    }
}

template <int LPR>
int launch_batch(const float *S_on, const float *S_tar, float div, int n_users, const float *W, const float *bias,
                 const int32_t *u, const int32_t *i, int B, float *dS, float *Xb, float *Gb, double *loss, float *contrib, int32_t *keys,
                 hipStream_t st) {
    constexpr int LD = 4 * LPR, GPW = kWave / LPR;
    const size_t lds = (size_t)(LD * LD + LD * (LD + 1)) * sizeof(float);
    if (lds > 64 * 1024)
        QREC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&buir_batch_kernel<LPR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = ((int64_t)B + 4 * GPW - 1) / (4 * GPW);
    if (blocks > 256) blocks = 256;          // every block stages the weights once
    hipLaunchKernelGGL((buir_batch_kernel<LPR>), dim3((unsigned)blocks), dim3(256), lds, st, S_on, S_tar, div, n_users, W, bias, u,
                       i, B, dS, Xb, Gb, loss, contrib, keys);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // namespace

extern "C" {

int qrec_buir_batch_loss_grad(const float *d_S_online, const float *d_S_target, float div, int32_t n_users, int32_t ld,
                              const float *d_W, const float *d_bias, const int32_t *d_u, const int32_t *d_i, int32_t B,
                              float *d_dS, float *d_X, float *d_dPre, double *d_loss, void *d_ordered_ws, int64_t ordered_ws_bytes,
                              void *stream) {
    QREC_REQUIRE(d_S_online && d_S_target && d_W && d_bias && d_dS && d_X && d_dPre && d_loss && B >= 0 && div != 0.f,
                 "qrec_buir_batch_loss_grad: bad argument");
    QREC_REQUIRE(B == 0 || (d_u && d_i), "qrec_buir_batch_loss_grad: null index array");
    if (B == 0) return QREC_OK;
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128, "qrec_buir_batch_loss_grad: row stride must be 32, 64 or 128 floats (got %d)", ld);
    hipStream_t st = as_stream(stream);
    OrderedScatterWs ow = {};
    if (d_ordered_ws) {
        const int rc = ordered_ws_carve(d_ordered_ws, ordered_ws_bytes, 2 * (int64_t)B, ld, &ow);
        if (rc != QREC_OK) return rc;
    }
    int rc = QREC_OK;
    switch (ld) {
        case 32: rc = launch_batch<8>(d_S_online, d_S_target, div, n_users, d_W, d_bias, d_u, d_i, B, d_dS, d_X, d_dPre, d_loss, ow.contrib, ow.keys, st); break;
        case 64: rc = launch_batch<16>(d_S_online, d_S_target, div, n_users, d_W, d_bias, d_u, d_i, B, d_dS, d_X, d_dPre, d_loss, ow.contrib, ow.keys, st); break;
        default: rc = launch_batch<32>(d_S_online, d_S_target, div, n_users, d_W, d_bias, d_u, d_i, B, d_dS, d_X, d_dPre, d_loss, ow.contrib, ow.keys, st); break;
    }
    if (rc != QREC_OK || !d_ordered_ws) return rc;
    return ordered_scatter_run(ow, 2 * (int64_t)B, ld, B, d_dS, st);      // slots [0, B): the users' online rows, [B, 2B): the items'
}

int qrec_buir_wgrad_scratch_bytes(int32_t ld, int64_t *bytes) {
    QREC_REQUIRE(bytes && ld > 0, "qrec_buir_wgrad_scratch_bytes: bad argument");
    *bytes = (int64_t)kWgradSlices * (ld + 1) * ld * sizeof(float);
    return QREC_OK;
}

int qrec_buir_wgrad(const float *d_X, const float *d_dPre, int32_t n_rows, int32_t ld, float *d_scratch, float *d_gW,
                    float *d_gb, void *stream) {
    QREC_REQUIRE(d_X && d_dPre && d_scratch && d_gW && d_gb && n_rows >= 0, "qrec_buir_wgrad: bad argument");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128, "qrec_buir_wgrad: row stride must be 32, 64 or 128 floats (got %d)", ld);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(buir_wgrad_kernel, dim3((unsigned)(ld + 1), kWgradSlices), dim3(256), 0, st, d_X, d_dPre, n_rows, ld, d_scratch);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(buir_wgrad_reduce_kernel, dim3((unsigned)(((ld + 1) * ld + 255) / 256)), dim3(256), 0, st, d_scratch, ld, d_gW, d_gb);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_ema_update(float *d_target, const float *d_online, float tau, int64_t n_elems, void *stream) {
    QREC_REQUIRE(d_target && d_online && n_elems >= 0 && n_elems % 4 == 0, "qrec_ema_update: bad argument");
    if (n_elems == 0) return QREC_OK;
    int64_t blocks = (n_elems / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_target, d_online, tau, n_elems / 4);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
