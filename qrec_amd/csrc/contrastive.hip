// SimGCL extras (model/ranking/SimGCL.py): the noise perturbation of the two augmented views
// (:29-38) and the InfoNCE contrastive loss with its gradients (:60-90).
//
//   perturb_kernel          emb += sign(emb) * l2_normalize(noise) * eps ; accum += emb
//   gather_normalize_kernel z = l2_normalize(S[row]/div) for the batch's unique rows, both views
//   exp_logits_kernel       ExT[b][a] = exp(z1[a].z2[b] / tau)               (MFMA f32 32x32x2)
//   row_stats_kernel        ttl[a] = sum_b Ex[a][b] (from per-tile partial sums);  loss += -log(exp(z1[a].z2[a]/tau) / ttl[a])
//   grad_z_kernel           dz1 = (P - I) z2 / tau ; dz2 = (P - I)^T z1 / tau , P = Ex / ttl   (MFMA, split-K x16)
//   normalize_bwd_kernel    d_out[row] += cl_rate * (dx1 + dx2),  dx = (dz - z (z.dz)) / |x|
//
// The n x n logits block (n <= batch size) is the only GEMM-shaped work: 3 products of
// 2 n^2 d FLOP; everything else streams rows.
#include "common.h"

using namespace qrec;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- Philox4x32-10 (same generator as the negative sampler) ---------------------------------
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

// One group of LPR lanes per row, float4 per lane (ld = 4*LPR).  V views are formed from the same source row in one pass.
//   noise[v] != nullptr : use the given U[0,1) numbers (parity tests inject TF-side noise).  An injected value may also carry the SIGN
//                         the perturbation is to use instead of sign(emb) -- sign() is discontinuous, and a test that follows a
//                         recorded reference run past an entry within rounding of zero feeds the recorded pattern (include/qrec_hip.h):
//                         v in [0, 1) natural;  +-(2 + u): magnitude u, sign forced to +-1;  4 + u: magnitude u, sign forced to 0
//   noise[v] == nullptr : Philox(counter = {row, lane, stream_lo, stream_hi}, key = seed) -> 4 x 24-bit uniforms
//   src == nullptr      : in place on emb[0] (V = 1)
//   assign              : sum[v] = emb_v instead of sum[v] += emb_v, and src_sum (if given) = the source row: the FIRST layer
//                         of the three encoders starts their layer sums, which therefore need no zero-fill
//   row_ids             : only the listed rows (the last layer of a training step is read at the batch's rows only)
struct PerturbViews {
    float *emb[2];
    const float *noise[2];
    uint64_t stream_id[2];
    float *sum[2];
};
template <int LPR, int V>
__global__ __launch_bounds__(256) void perturb_kernel(PerturbViews pv, const float *__restrict__ src, int64_t n_rows,
                                                      int d, float eps, uint64_t seed, int assign, float *__restrict__ src_sum,
                                                      const int32_t *__restrict__ row_ids, const int32_t *__restrict__ n_ids,
                                                      int64_t philox_row0) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    const int64_t n_groups = (int64_t)gridDim.x * 4 * GPW;
    const int64_t n_todo = row_ids ? *n_ids : n_rows;
    for (int64_t k = gid; k < n_todo; k += n_groups) {
        const int64_t row = row_ids ? (int64_t)row_ids[k] : k;
        const int64_t off = row * (4 * LPR) + 4 * r;
        const f32x4 x = *reinterpret_cast<const f32x4 *>((src ? src : pv.emb[0]) + off);
        if (src_sum) *reinterpret_cast<f32x4 *>(src_sum + off) = x;
#pragma unroll
        for (int v = 0; v < V; v++) {
            f32x4 nz;
            f32x4 forced = {2.f, 2.f, 2.f, 2.f};          // 2 = not forced
            if (pv.noise[v]) {
                nz = *reinterpret_cast<const f32x4 *>(pv.noise[v] + off);
                auto mag = [](float t) { const float a = fabsf(t); return a >= 4.f ? a - 4.f : (a >= 2.f ? a - 2.f : t); };
                auto frc = [](float t) { const float a = fabsf(t); return a >= 4.f ? 0.f : (a >= 2.f ? (t > 0.f ? 1.f : -1.f) : 2.f); };
                forced.x = frc(nz.x); forced.y = frc(nz.y); forced.z = frc(nz.z); forced.w = frc(nz.w);
                nz.x = mag(nz.x); nz.y = mag(nz.y); nz.z = mag(nz.z); nz.w = mag(nz.w);
            } else {
                const uint64_t sid = pv.stream_id[v];
                const int64_t grow = row + philox_row0;      // the table row this block row stands for (row-partitioned tables)
                uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32) ^ ((uint32_t)r << 8), (uint32_t)sid, (uint32_t)(sid >> 32)};
                philox10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
                nz.x = (float)(c[0] >> 8) * 0x1p-24f; nz.y = (float)(c[1] >> 8) * 0x1p-24f;
                nz.z = (float)(c[2] >> 8) * 0x1p-24f; nz.w = (float)(c[3] >> 8) * 0x1p-24f;
            }
            if (4 * r + 0 >= d) nz.x = 0.f;
            if (4 * r + 1 >= d) nz.y = 0.f;
            if (4 * r + 2 >= d) nz.z = 0.f;
            if (4 * r + 3 >= d) nz.w = 0.f;
            float ss = nz.x * nz.x + nz.y * nz.y + nz.z * nz.z + nz.w * nz.w;
            ss = row_allreduce_sum<LPR>(ss);
            const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));     // tf.nn.l2_normalize epsilon
            auto sgn = [](float t, float f) { return f != 2.f ? f : (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)); };
            f32x4 e = x;
            e.x += sgn(e.x, forced.x) * (nz.x * inv) * eps; e.y += sgn(e.y, forced.y) * (nz.y * inv) * eps;
            e.z += sgn(e.z, forced.z) * (nz.z * inv) * eps; e.w += sgn(e.w, forced.w) * (nz.w * inv) * eps;
            *reinterpret_cast<f32x4 *>(pv.emb[v] + off) = e;
            if (pv.sum[v]) {
                f32x4 t = e;
                if (!assign) t = t + *reinterpret_cast<const f32x4 *>(pv.sum[v] + off);
                *reinterpret_cast<f32x4 *>(pv.sum[v] + off) = t;
            }
        }
    }
}

// z[k] = l2_normalize(S[rows[k]] / div) for both views; r = 1/|x|; dotp = z1.z2
template <int LPR>
__global__ __launch_bounds__(256) void gather_normalize_kernel(const float *__restrict__ S1, const float *__restrict__ S2,
                                                               float div, const int32_t *__restrict__ rows, int n, int n_pad,
                                                               float *__restrict__ z1, float *__restrict__ z2,
                                                               float *__restrict__ r1, float *__restrict__ r2,
                                                               float *__restrict__ dotp) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    if (k >= n_pad) return;
    if (k >= n) {   // pad rows: zeros, so that the MFMA kernels can load whole tiles unconditionally
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(z1 + k * (4 * LPR) + 4 * r) = zero; *reinterpret_cast<f32x4 *>(z2 + k * (4 * LPR) + 4 * r) = zero;
        if (r == 0) { r1[k] = 0.f; r2[k] = 0.f; dotp[k] = 0.f; }
        return;
    }
    const int64_t src = (int64_t)rows[k] * (4 * LPR) + 4 * r, dst = k * (4 * LPR) + 4 * r;
    f32x4 a = *reinterpret_cast<const f32x4 *>(S1 + src), b = *reinterpret_cast<const f32x4 *>(S2 + src);
    a.x /= div; a.y /= div; a.z /= div; a.w /= div; b.x /= div; b.y /= div; b.z /= div; b.w /= div;
    float sa = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w, sb = b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    sa = row_allreduce_sum<LPR>(sa); sb = row_allreduce_sum<LPR>(sb);
    const float ia = 1.0f / sqrtf(fmaxf(sa, 1e-12f)), ib = 1.0f / sqrtf(fmaxf(sb, 1e-12f));
    a = a * ia; b = b * ib;
    float dp = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    dp = row_allreduce_sum<LPR>(dp);
    *reinterpret_cast<f32x4 *>(z1 + dst) = a; *reinterpret_cast<f32x4 *>(z2 + dst) = b;
    if (r == 0) { r1[k] = ia; r2[k] = ib; dotp[k] = dp; }
}

// ExT[b][a] = exp(z1[a].z2[b] * inv_tau): one wavefront per 32x32 tile, f32 MFMA.
// lane l: row index r = l&31, k-slot h = l>>5 owns columns [64c+32h, 64c+32h+32) of chunk c: 32 consecutive floats of
// ITS row for each operand, fetched as 8 unconditional float4 loads (z1, z2 are zero-padded to n_pad rows).
__global__ __launch_bounds__(256) void exp_logits_kernel(const float *__restrict__ z1, const float *__restrict__ z2,
                                                         int n, int n_pad, int ld, float inv_tau,
                                                         float *__restrict__ ExT, float *__restrict__ Ex,
                                                         float *__restrict__ psum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int a0 = blockIdx.x * 32, b0 = (blockIdx.y * 4 + wave) * 32;
    if (b0 >= n_pad) return;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.f;
    for (int c = 0; c < ld; c += 64) {
        const int col0 = c + 32 * h;
        const bool kv = col0 < ld;                                  // ld = 32: the second k-slot has no columns
        const f32x4 *pb = reinterpret_cast<const f32x4 *>(z2 + (int64_t)(b0 + r) * ld + (kv ? col0 : 0));   // A: rows b
        const f32x4 *pa = reinterpret_cast<const f32x4 *>(z1 + (int64_t)(a0 + r) * ld + (kv ? col0 : 0));   // B: cols a
        f32x4 vb[8], va[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { vb[q] = pb[q]; va[q] = pa[q]; }
        const float keep = kv ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[q].x * keep, va[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[q].y * keep, va[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[q].z * keep, va[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[q].w * keep, va[q].w, acc, 0, 0, 0);
        }
    }
    // C/D: col = lane&31 -> a, row = (q&3) + 8*(q>>2) + 4*h -> b
    // Both orientations are stored: each gradient product then reads its A operand along its own rows (float4).
    float part = 0.f;   // sum over this tile's 32 b's for column a (fixed order: deterministic)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
        const int a = a0 + r, bq = b0 + 8 * g4 + 4 * h;
        f32x4 e4;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int b = bq + t;
            const float e = (a < n && b < n) ? expf(acc[4 * g4 + t] * inv_tau) : 0.f;
            ExT[(int64_t)b * n_pad + a] = e;
            e4[t] = e;
            part += e;
        }
        *reinterpret_cast<f32x4 *>(Ex + (int64_t)a * n_pad + bq) = e4;
    }
    part += __shfl_xor(part, 32, kWave);
    if (h == 0) psum[(int64_t)(b0 / 32) * n_pad + a0 + r] = part;
}

// ttl[a] = sum over b-tiles of psum[tile][a];  loss_a = -log(exp(dotp[a]*inv_tau) / ttl[a])
// One thread per column a, the tiles in order (deterministic): consecutive threads read consecutive floats of a tile's row of psum, so the
// n_pad / 32 loads of a thread are coalesced across the wavefront and independent of each other.  (Rounds 1-4 gave a column to 8 lanes, each
// walking every 8th tile -- 64 different 8 KB-strided lines per load instruction: 13 us for 0.5 MB at n = 2048; round 5.)
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ psum, int n_parts, int n, int n_pad,
                                                        const float *__restrict__ dotp, float inv_tau,
                                                        float *__restrict__ inv_ttl, double *__restrict__ loss_out) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    double l = 0.0;
    if (a < n_pad) {
        float ttl = 0.f;
        if (a < n) {
            const int tiles = n_parts;
            int t = 0;
            for (; t + 8 <= tiles; t += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = psum[(int64_t)(t + q) * n_pad + a];
#pragma unroll
                for (int q = 0; q < 8; q++) ttl += v[q];
            }
            for (; t < tiles; t++) ttl += psum[(int64_t)t * n_pad + a];
            inv_ttl[a] = 1.0f / ttl;
            l = (double)(-logf(expf(dotp[a] * inv_tau) / ttl));
        } else {
            inv_ttl[a] = 0.f;       // pad entries are read (and multiplied by 0) by grad_z_kernel
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m, kWave);
    __shared__ double s_l[4];
    if ((threadIdx.x & 63) == 0) s_l[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
        if (tot != 0.0) atomicAdd(loss_out, tot);      // one fp64 atomic per block
    }
}

// G[a][b] = ExT[b][a]*inv_ttl[a] - (a==b).
// MODE 0: out[a][:] = inv_tau * sum_b G[a][b] z[b][:]      (dz1, z = z2)   A[i=a][k=b] = G[a][b]
// MODE 1: out[b][:] = inv_tau * sum_a G[a][b] z[a][:]      (dz2, z = z1)   A[i=b][k=a] = G[a][b]
// one wavefront per (32 output rows) x (32 output columns); K = n in chunks of 64 (2 slots x 32).
constexpr int kSplitK = 16;   // K = n is cut into 16 slices -> 16x more wavefronts; partials summed by the consumer

template <int MODE>
__global__ __launch_bounds__(256) void grad_z_kernel(const float *__restrict__ EA, const float *__restrict__ inv_ttl,
                                                     const float *__restrict__ z, int n, int n_pad, int ld, float inv_tau,
                                                     float diag, float *__restrict__ out_parts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int ks = blockIdx.z * 4 + wave;                 // K slice
    const int i = i0 + r;
    const int k_per = ((n_pad / 64 + kSplitK - 1) / kSplitK) * 64;
    const int c_beg = ks * k_per, c_end = (c_beg + k_per) < n_pad ? (c_beg + k_per) : n_pad;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.f;
    // all loads are unconditional: ExT, inv_ttl and z are zero in their pad rows/entries
    const float my_inv = MODE == 0 ? inv_ttl[i] : 0.f;
    const float *zcol = z + j0 + r;
    for (int c = c_beg; c < c_end; c += 64) {
        const int k0 = c + 32 * h;
        // A[i][k] = EA[i][k] * scale - (i == k), EA = Ex (MODE 0, scale = 1/ttl[i]) or ExT (MODE 1, scale = 1/ttl[k]):
        // 32 consecutive floats of the lane's own row -> 8 float4 loads
        const f32x4 *pe = reinterpret_cast<const f32x4 *>(EA + (int64_t)i * n_pad + k0);
        const f32x4 *pt = reinterpret_cast<const f32x4 *>(inv_ttl + k0);
        const float dgl = i < n ? diag : 0.f;      // diag = 1: the positive is the row's own column (SimGCL); 0: none here (SEPT)
        // a chunk's operands first (8 float4 of the lane's E row, 32 coalesced dwords of z), then its 32 MFMAs: with
        // one load issued per MFMA the compiler keeps it a single MFMA ahead and the wavefront sits in load latency
        float ga[32], zb[32];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const f32x4 e = pe[q];
            f32x4 t = {my_inv, my_inv, my_inv, my_inv};
            if (MODE == 1) t = pt[q];
            const int k = k0 + 4 * q;
            ga[4 * q + 0] = e.x * t.x - (k + 0 == i ? dgl : 0.f); ga[4 * q + 1] = e.y * t.y - (k + 1 == i ? dgl : 0.f);
            ga[4 * q + 2] = e.z * t.z - (k + 2 == i ? dgl : 0.f); ga[4 * q + 3] = e.w * t.w - (k + 3 == i ? dgl : 0.f);
        }
#pragma unroll
        for (int s = 0; s < 32; s++) zb[s] = zcol[(int64_t)(k0 + s) * ld];
#pragma unroll
        for (int s = 0; s < 32; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[s], zb[s], acc, 0, 0, 0);
    }
    float *out = out_parts + (int64_t)ks * n_pad * ld;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h, col = j0 + r;
        if (row < n_pad && col < ld) out[(int64_t)row * ld + col] = acc[q] * inv_tau;
    }
}

// (Round 5, built, measured and removed: InfoNCE WITHOUT the n x n block -- every consumer recomputes its 32 x 32 tile of logits from the two
// 0.5 MB operand tables and keeps it in registers; the C layout of v_mfma_f32_32x32x2f32 is the B-operand layout of the next MFMA when the
// reduction walks the tile's rows in that order, so the tile never leaves the registers.  Correct on the first run (same tests, same bounds),
// and not faster at n = 2,048: row sums 19.0 us + the two gradient products 28.2 + 24.8 us = 72 us against exp_logits 21.4 + grad_z 20.7 +
// 20.4 = 62.5 us -- 96 dependent fp32 MFMAs per tile at 64 clocks each are 2.6 us per tile and wavefront, one wavefront per SIMD, and the
// operand loads of the next tile could not be hidden behind them (double-buffered: 29 + 38 us, 292 registers).  At K = d = 64 the fp32 MFMA
// rate, not the 33 MB of logits, is the floor of this formulation.  profiles/r05_simgcl_flash_nce*_kernel_stats.txt; the kernels are in
// git history (commit "auto never picks the deferred-negatives schedule ...").)
// d_out[rows[k]] += scale * ( (dz1 - z1 (z1.dz1)) r1 + (dz2 - z2 (z2.dz2)) r2 )
template <int LPR>
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float *__restrict__ z1, const float *__restrict__ z2,
                                                            const float *__restrict__ dz1, const float *__restrict__ dz2,
                                                            const float *__restrict__ r1, const float *__restrict__ r2,
                                                            const int32_t *__restrict__ rows, int n, int64_t part_stride,
                                                            float scale, float *__restrict__ d_out, float *__restrict__ d_out2) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    if (k >= n) return;
    const int64_t src = k * (4 * LPR) + 4 * r, dst = (int64_t)rows[k] * (4 * LPR) + 4 * r;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(z1 + src), b = *reinterpret_cast<const f32x4 *>(z2 + src);
    f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da;
#pragma unroll
    for (int ks = 0; ks < kSplitK; ks++) {
        da = da + *reinterpret_cast<const f32x4 *>(dz1 + (int64_t)ks * part_stride + src);
        db = db + *reinterpret_cast<const f32x4 *>(dz2 + (int64_t)ks * part_stride + src);
    }
    float pa = a.x * da.x + a.y * da.y + a.z * da.z + a.w * da.w, pb = b.x * db.x + b.y * db.y + b.z * db.z + b.w * db.w;
    pa = row_allreduce_sum<LPR>(pa); pb = row_allreduce_sum<LPR>(pb);
    const f32x4 dxa = (da - a * pa) * r1[k], dxb = (db - b * pb) * r2[k];
    f32x4 o = *reinterpret_cast<const f32x4 *>(d_out + dst);     // rows[] are distinct: plain read-modify-write
    if (d_out2) {   // the two views back-propagate through different operators (SGL): separate accumulators
        f32x4 o2 = *reinterpret_cast<const f32x4 *>(d_out2 + dst);
        o = o + scale * dxa; o2 = o2 + scale * dxb;
        *reinterpret_cast<f32x4 *>(d_out2 + dst) = o2;
    } else {
        o = o + scale * dxa + scale * dxb;
    }
    *reinterpret_cast<f32x4 *>(d_out + dst) = o;
}

template <int LPR>
int run_info_nce(const float *S1, const float *S2, float div, const int32_t *rows, int n, int ld, float tau,
                 float cl_rate, float *ws, float *d_out, float *d_out2, double *loss, hipStream_t st) {
    constexpr int GPW = kWave / LPR;
    const int n_pad = (n + 63) / 64 * 64;
    const int64_t tab = (int64_t)n_pad * ld;
    float *z1 = ws, *z2 = z1 + tab, *dz1 = z2 + tab, *dz2 = dz1 + kSplitK * tab;
    float *r1 = dz2 + kSplitK * tab, *r2 = r1 + n_pad, *dotp = r2 + n_pad, *inv_ttl = dotp + n_pad;
    float *psum = inv_ttl + n_pad;                       // [n_pad/32][n_pad]
    float *ExT = psum + (int64_t)(n_pad / 32) * n_pad;   // [n_pad][n_pad]
    float *Ex = ExT + (int64_t)n_pad * n_pad;            // [n_pad][n_pad], the other orientation
    const float inv_tau = 1.0f / tau;
    const unsigned row_blocks = (unsigned)((n + 4 * GPW - 1) / (4 * GPW)), pad_blocks = (unsigned)((n_pad + 4 * GPW - 1) / (4 * GPW));
    hipLaunchKernelGGL((gather_normalize_kernel<LPR>), dim3(pad_blocks), dim3(256), 0, st, S1, S2, div, rows, n, n_pad, z1, z2, r1, r2, dotp);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(exp_logits_kernel, dim3((unsigned)(n_pad / 32), (unsigned)((n_pad / 32 + 3) / 4)), dim3(256), 0, st,
                       z1, z2, n, n_pad, ld, inv_tau, ExT, Ex, psum);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, psum, n_pad / 32, n, n_pad, dotp, inv_tau, inv_ttl, loss);
    QREC_LAUNCH_CHECK();
    const dim3 gg((unsigned)(n_pad / 32), (unsigned)((ld + 31) / 32), (unsigned)(kSplitK / 4));
    hipLaunchKernelGGL((grad_z_kernel<0>), gg, dim3(256), 0, st, Ex, inv_ttl, z2, n, n_pad, ld, inv_tau, 1.f, dz1);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL((grad_z_kernel<1>), gg, dim3(256), 0, st, ExT, inv_ttl, z1, n, n_pad, ld, inv_tau, 1.f, dz2);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL((normalize_bwd_kernel<LPR>), dim3(row_blocks), dim3(256), 0, st, z1, z2, dz1, dz2, r1, r2, rows, n, tab, cl_rate, d_out, d_out2);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

// =============================================================================================
// SEPT (model/ranking/SEPT.py:214-262): tri-training pseudo labels and the neighbour-discrimination loss on the
// batch's unique users.  z_v = l2_normalize(S_v[rows]) for the three encoders (friend, sharing, preference view),
// a = l2_normalize(S_aug[rows]).  Reuses gather_normalize / exp_logits / grad_z above.
//   row_total_kernel      ttl[i] = sum_j Ex[i][j] from the per-tile partial sums
//   topk_pair_kernel      labels[i][0..k) = top_k((softmax_p[i] + softmax_q[i]) / 2), equal scores in index order
//   sept_pos_kernel       pos[i] = sum_t Ex[i][labels[i][t]];  loss += -log(pos[i] / ttl[i])
//   sept_sparse_kernel    the positives' part of the gradient: dzc[i] = -(1/tau) sum_t (Ex[i][l]/pos[i]) a[l],
//                         dac[l] -= (1/tau) (Ex[i][l]/pos[i]) z[i]   (l = labels[i][t]; atomics: rows share positives)
//   sept_normalize_bwd    d_out[rows[i]] += scale * (dz - z (z.dz)) * r,  dz = sum of the split-K parts (+ sparse part)
// =============================================================================================
__global__ __launch_bounds__(256) void row_total_kernel(const float *__restrict__ psum, int n, int n_pad, float *__restrict__ ttl,
                                                        float *__restrict__ inv_ttl) {
    const int a = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, t0 = threadIdx.x & 7;
    float s = 0.f;
    if (a < n)
        for (int t = t0; t < n_pad / 32; t += 8) s += psum[(int64_t)t * n_pad + a];
    s += __shfl_xor(s, 1, kWave); s += __shfl_xor(s, 2, kWave); s += __shfl_xor(s, 4, kWave);
    if (t0 == 0 && a < n_pad) {
        ttl[a] = a < n ? s : 1.f;
        if (inv_ttl) inv_ttl[a] = a < n ? 1.0f / s : 0.f;
    }
}

// One wavefront per row.  The row's n averaged scores are formed once and kept in registers (EPL per lane, column
// j = e * 64 + lane); a selection round is a register scan + a (value, index) butterfly, the winner's slot is then
// closed.  Ties go to the lower column, as tf.math.top_k's.  EPL = 0: rows longer than 4096 columns re-read the two
// score rows from L2 in every round instead.
template <int EPL>
__global__ __launch_bounds__(256) void topk_pair_kernel(const float *__restrict__ Ep, const float *__restrict__ tp,
                                                        const float *__restrict__ Eq, const float *__restrict__ tq,
                                                        int n, int n_pad, int k, int32_t *__restrict__ labels) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *p = Ep + (int64_t)i * n_pad, *q = Eq + (int64_t)i * n_pad;
    const float sp = tp[i], sq = tq[i];
    auto score = [&](int j) { return (p[j] / sp + q[j] / sq) / 2.0f; };      // (prob1 + prob2) / 2, SEPT.py:229
    auto wave_best = [&](float &bv, int &bj) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(bv, m, kWave);
            const int oj = __shfl_xor(bj, m, kWave);
            if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
        }
    };
    if constexpr (EPL > 0) {
        float v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int j = e * kWave + lane;
            v[e] = j < n ? score(j) : -INFINITY;
        }
        for (int t = 0; t < k; t++) {
            float bv = -INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int e = 0; e < EPL; e++)
                if (v[e] > bv) { bv = v[e]; bj = e * kWave + lane; }          // ascending columns: the first of equals stays
            wave_best(bv, bj);
            if (lane == 0) labels[(int64_t)i * k + t] = bj;
#pragma unroll
            for (int e = 0; e < EPL; e++)
                if (e * kWave + lane == bj) v[e] = -INFINITY;
        }
    } else {
        float last_v = INFINITY;
        int last_j = -1;
        for (int t = 0; t < k; t++) {
            float bv = -INFINITY;
            int bj = 0x7fffffff;
            for (int j = lane; j < n; j += kWave) {
                const float v = score(j);
                const bool open = v < last_v || (v == last_v && j > last_j);   // not selected in an earlier round
                if (open && (v > bv || (v == bv && j < bj))) { bv = v; bj = j; }
            }
            wave_best(bv, bj);
            if (lane == 0) labels[(int64_t)i * k + t] = bj;
            last_v = bv; last_j = bj;
        }
    }
}

__global__ __launch_bounds__(256) void sept_pos_kernel(const float *__restrict__ Ex, const float *__restrict__ ttl,
                                                       const int32_t *__restrict__ labels, int n, int n_pad, int k,
                                                       float *__restrict__ inv_pos, double *__restrict__ loss_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double l = 0.0;
    if (i < n) {
        float ps = 0.f;
        for (int t = 0; t < k; t++) ps += Ex[(int64_t)i * n_pad + labels[(int64_t)i * k + t]];
        inv_pos[i] = 1.0f / ps;
        l = (double)(-logf(ps / ttl[i]));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m, kWave);
    if ((threadIdx.x & 63) == 0 && l != 0.0) atomicAdd(loss_out, l);
}

template <int LPR>
__global__ __launch_bounds__(256) void sept_sparse_kernel(const float *__restrict__ Ex, const float *__restrict__ inv_pos,
                                                          const int32_t *__restrict__ labels, const float *__restrict__ z,
                                                          const float *__restrict__ a, int n, int n_pad, int k, float inv_tau,
                                                          float *__restrict__ dzc, float *__restrict__ dac, float *__restrict__ contrib,
                                                          int32_t *__restrict__ keys) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t i = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    if (i >= n) return;
    const f32x4 zi = *reinterpret_cast<const f32x4 *>(z + i * (4 * LPR) + 4 * r);
    const float ip = inv_pos[i];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < k; t++) {
        const int l = labels[i * k + t];
        const float c = Ex[i * n_pad + l] * ip * inv_tau;
        const f32x4 al = *reinterpret_cast<const f32x4 *>(a + (int64_t)l * (4 * LPR) + 4 * r);
        acc = acc - c * al;
        if (contrib) {      // parity mode: slot (i, t) of the ordered-scatter workspace; ordered.hip adds a positive's rows in (i, t) order
            *reinterpret_cast<f32x4 *>(contrib + (i * k + t) * (4 * LPR) + 4 * r) = -c * zi;
            if (r == 0) keys[i * k + t] = l;
        } else {
            float *dst = dac + (int64_t)l * (4 * LPR) + 4 * r;
            unsafeAtomicAdd(dst + 0, -c * zi.x); unsafeAtomicAdd(dst + 1, -c * zi.y);
            unsafeAtomicAdd(dst + 2, -c * zi.z); unsafeAtomicAdd(dst + 3, -c * zi.w);
        }
    }
    *reinterpret_cast<f32x4 *>(dzc + i * (4 * LPR) + 4 * r) = acc;
}

// dz = sum over n_parts split-K partials (part_stride apart) + extra;  d_out[rows[i]] += scale * (dz - z (z.dz)) * r
template <int LPR>
__global__ __launch_bounds__(256) void sept_normalize_bwd_kernel(const float *__restrict__ z, const float *__restrict__ parts,
                                                                 int n_parts, int64_t part_stride, const float *__restrict__ extra,
                                                                 const float *__restrict__ rinv, const int32_t *__restrict__ rows,
                                                                 int n, float scale, float *__restrict__ d_out) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t i = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    if (i >= n) return;
    const int64_t src = i * (4 * LPR) + 4 * r, dst = (int64_t)rows[i] * (4 * LPR) + 4 * r;
    const f32x4 zi = *reinterpret_cast<const f32x4 *>(z + src);
    f32x4 d = *reinterpret_cast<const f32x4 *>(extra + src);
    for (int p = 0; p < n_parts; p++) d = d + *reinterpret_cast<const f32x4 *>(parts + (int64_t)p * part_stride + src);
    float dot = zi.x * d.x + zi.y * d.y + zi.z * d.z + zi.w * d.w;
    dot = row_allreduce_sum<LPR>(dot);
    const f32x4 dx = (d - zi * dot) * rinv[i];
    f32x4 o = *reinterpret_cast<const f32x4 *>(d_out + dst);          // rows[] are distinct
    o = o + scale * dx;
    *reinterpret_cast<f32x4 *>(d_out + dst) = o;
}

inline int64_t sept_ws_floats(int64_t n, int64_t ld, int64_t k) {
    const int64_t n_pad = (n + 63) / 64 * 64, tab = n_pad * ld;
    // z x4, dzc, dac, dz parts (1 set), da parts (3 sets), r x4 + dotp scratch, ttl1 x3, ttl, inv_ttl, inv_pos, psum, Ex1 x3, Ex, ExT, labels
    return 6 * tab + 4 * kSplitK * tab + 11 * n_pad + (n_pad / 32) * n_pad + 5 * n_pad * n_pad + 3 * n_pad * k;
}

template <int LPR>
int run_sept_ssl(const float *const S[4], const int32_t *rows, int n, int ld, int k, float ss_rate, float *ws,
                 float *const dS[4], double *loss, int32_t *labels_out, const OrderedScatterWs *ow, hipStream_t st) {
    constexpr int GPW = kWave / LPR;
    const int n_pad = (n + 63) / 64 * 64;
    const int64_t tab = (int64_t)n_pad * ld, sq = (int64_t)n_pad * n_pad;
    float *z[4]; float *p = ws;
    for (int v = 0; v < 4; v++) { z[v] = p; p += tab; }
    float *dzc = p; p += tab;
    float *dac = p; p += tab;
    float *dzp = p; p += kSplitK * tab;
    float *dap = p; p += 3 * kSplitK * tab;
    float *rinv[4];
    for (int v = 0; v < 4; v++) { rinv[v] = p; p += n_pad; }
    float *dotp = p; p += n_pad;
    float *ttl1[3];
    for (int v = 0; v < 3; v++) { ttl1[v] = p; p += n_pad; }
    float *ttl = p; p += n_pad;
    float *inv_ttl = p; p += n_pad;
    float *inv_pos = p; p += n_pad;
    float *psum = p; p += (int64_t)(n_pad / 32) * n_pad;
    float *Ex1[3];
    for (int v = 0; v < 3; v++) { Ex1[v] = p; p += sq; }
    float *Ex = p; p += sq;
    float *ExT = p; p += sq;
    int32_t *labels = reinterpret_cast<int32_t *>(p);                       // [3][n][k]
    const float inv_tau = 10.0f;                                            // tau = 0.1, SEPT.py:245-246
    const unsigned row_blocks = (unsigned)((n + 4 * GPW - 1) / (4 * GPW)), pad_blocks = (unsigned)((n_pad + 4 * GPW - 1) / (4 * GPW));
    const dim3 eg((unsigned)(n_pad / 32), (unsigned)((n_pad / 32 + 3) / 4));
    const dim3 gg((unsigned)(n_pad / 32), (unsigned)((ld + 31) / 32), (unsigned)(kSplitK / 4));
    const unsigned stat_blocks = (unsigned)((n_pad * 8 + 255) / 256);
    // z_f, z_h | z_e, a   (S[0] friend, S[1] sharing, S[2] preference, S[3] augmented view)
    hipLaunchKernelGGL((gather_normalize_kernel<LPR>), dim3(pad_blocks), dim3(256), 0, st, S[0], S[1], 1.0f, rows, n, n_pad, z[0], z[1], rinv[0], rinv[1], dotp);
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL((gather_normalize_kernel<LPR>), dim3(pad_blocks), dim3(256), 0, st, S[2], S[3], 1.0f, rows, n, n_pad, z[2], z[3], rinv[2], rinv[3], dotp);
    QREC_LAUNCH_CHECK();
    // label_prediction (SEPT.py:214-224): softmax rows of z_v a^T -> exp and row totals, tau = 1
    for (int v = 0; v < 3; v++) {
        hipLaunchKernelGGL(exp_logits_kernel, eg, dim3(256), 0, st, z[v], z[3], n, n_pad, ld, 1.0f, ExT, Ex1[v], psum);
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL(row_total_kernel, dim3(stat_blocks), dim3(256), 0, st, psum, n, n_pad, ttl1[v], (float *)nullptr);
        QREC_LAUNCH_CHECK();
    }
    // pseudo labels of encoder v = top-k of the OTHER two encoders' averaged predictions (SEPT.py:258-260)
    const int other[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    for (int v = 0; v < 3; v++) {
#define QREC_TOPK(EPL)                                                                                                        \
    hipLaunchKernelGGL((topk_pair_kernel<EPL>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, Ex1[other[v][0]], ttl1[other[v][0]], \
                       Ex1[other[v][1]], ttl1[other[v][1]], n, n_pad, k, labels + (int64_t)v * n * k)
        if (n_pad <= 512) QREC_TOPK(8);
        else if (n_pad <= 1024) QREC_TOPK(16);
        else if (n_pad <= 2048) QREC_TOPK(32);
        else if (n_pad <= 4096) QREC_TOPK(64);
        else QREC_TOPK(0);
#undef QREC_TOPK
        QREC_LAUNCH_CHECK();
    }
    if (labels_out) QREC_HIP_CHECK(hipMemcpyAsync(labels_out, labels, sizeof(int32_t) * 3 * (size_t)n * k, hipMemcpyDeviceToDevice, st));
    QREC_HIP_CHECK(hipMemsetAsync(dac, 0, sizeof(float) * tab, st));
    // neighbor_discrimination per encoder (SEPT.py:233-248), tau = 0.1
    for (int v = 0; v < 3; v++) {
        const int32_t *lab = labels + (int64_t)v * n * k;
        hipLaunchKernelGGL(exp_logits_kernel, eg, dim3(256), 0, st, z[v], z[3], n, n_pad, ld, inv_tau, ExT, Ex, psum);
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL(row_total_kernel, dim3(stat_blocks), dim3(256), 0, st, psum, n, n_pad, ttl, inv_ttl);
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL(sept_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Ex, ttl, lab, n, n_pad, k, inv_pos, loss);
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL((grad_z_kernel<0>), gg, dim3(256), 0, st, Ex, inv_ttl, z[3], n, n_pad, ld, inv_tau, 0.f, dzp);
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL((grad_z_kernel<1>), gg, dim3(256), 0, st, ExT, inv_ttl, z[v], n, n_pad, ld, inv_tau, 0.f, dap + (int64_t)v * kSplitK * tab);
        QREC_LAUNCH_CHECK();
        hipLaunchKernelGGL((sept_sparse_kernel<LPR>), dim3(row_blocks), dim3(256), 0, st, Ex, inv_pos, lab, z[v], z[3], n, n_pad, k, inv_tau, dzc, dac,
                           ow ? ow->contrib : (float *)nullptr, ow ? ow->keys : (int32_t *)nullptr);
        QREC_LAUNCH_CHECK();
        if (ow) {
            const int rc = ordered_scatter_run(*ow, (int64_t)n * k, ld, 0, dac, st);
            if (rc != QREC_OK) return rc;
        }
        hipLaunchKernelGGL((sept_normalize_bwd_kernel<LPR>), dim3(row_blocks), dim3(256), 0, st, z[v], dzp, kSplitK, tab, dzc, rinv[v], rows, n, ss_rate, dS[v]);
        QREC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((sept_normalize_bwd_kernel<LPR>), dim3(row_blocks), dim3(256), 0, st, z[3], dap, 3 * kSplitK, tab, dac, rinv[3], rows, n, ss_rate, dS[3]);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // namespace

namespace {
template <int V>
int launch_perturb(const PerturbViews &pv, const float *d_src, int64_t n_rows, int32_t d, int32_t ld, float eps, uint64_t seed,
                   int assign, float *d_src_sum, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                   int64_t philox_row0, hipStream_t st) {
    const int64_t work_rows = d_row_ids ? max_row_ids : n_rows;
    if (work_rows == 0) return QREC_OK;
    int64_t blocks;
#define QREC_PT(LPR)                                                                                              \
    blocks = (work_rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 2048) blocks = 2048;                 \
    hipLaunchKernelGGL((perturb_kernel<LPR, V>), dim3((unsigned)blocks), dim3(256), 0, st, pv, d_src, n_rows, d, eps, seed, \
                       assign, d_src_sum, d_row_ids, d_n_row_ids, philox_row0)
    switch (ld) {
        case 32: QREC_PT(8); break;
        case 64: QREC_PT(16); break;
        case 128: QREC_PT(32); break;
        case 256: QREC_PT(64); break;
        default: set_error("qrec_perturb_rows: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_PT
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}
}  // namespace

extern "C" {

int qrec_perturb_rows(float *d_emb, const float *d_src, int64_t n_rows, int32_t d, int32_t ld, float eps,
                      const float *d_noise, uint64_t seed, uint64_t stream_id, float *d_accum, const int32_t *d_row_ids,
                      const int32_t *d_n_row_ids, int32_t max_row_ids, int64_t philox_row0, void *stream) {
    QREC_REQUIRE(d_emb && n_rows >= 0 && d >= 1 && ld >= d, "qrec_perturb_rows: bad argument");
    QREC_REQUIRE(!d_row_ids || (d_n_row_ids && max_row_ids >= 0), "qrec_perturb_rows: a row subset needs its count and a bound");
    PerturbViews pv = {{d_emb, nullptr}, {d_noise, nullptr}, {stream_id, 0}, {d_accum, nullptr}};
    return launch_perturb<1>(pv, d_src, n_rows, d, ld, eps, seed, 0, nullptr, d_row_ids, d_n_row_ids, max_row_ids, philox_row0, as_stream(stream));
}

int qrec_perturb_two_views(const float *d_src, float *d_emb1, float *d_emb2, int64_t n_rows, int32_t d, int32_t ld, float eps,
                           const float *d_noise1, const float *d_noise2, uint64_t seed, uint64_t stream_id1, uint64_t stream_id2,
                           float *d_sum1, float *d_sum2, float *d_src_sum, const int32_t *d_row_ids, const int32_t *d_n_row_ids,
                           int32_t max_row_ids, int64_t philox_row0, void *stream) {
    QREC_REQUIRE(d_src && d_emb1 && d_emb2 && d_emb1 != d_emb2 && n_rows >= 0 && d >= 1 && ld >= d, "qrec_perturb_two_views: bad argument");
    QREC_REQUIRE(!d_row_ids || (d_n_row_ids && max_row_ids >= 0), "qrec_perturb_two_views: a row subset needs its count and a bound");
    PerturbViews pv = {{d_emb1, d_emb2}, {d_noise1, d_noise2}, {stream_id1, stream_id2}, {d_sum1, d_sum2}};
    return launch_perturb<2>(pv, d_src, n_rows, d, ld, eps, seed, 1, d_src_sum, d_row_ids, d_n_row_ids, max_row_ids, philox_row0, as_stream(stream));
}

int qrec_info_nce_workspace_bytes(int32_t n, int32_t ld, int64_t *bytes) {
    QREC_REQUIRE(bytes && n >= 0 && ld > 0, "qrec_info_nce_workspace_bytes: bad argument");
    const int64_t n_pad = ((int64_t)n + 63) / 64 * 64;
    *bytes = 4 * ((2 + 2 * 16) * n_pad * ld + 4 * n_pad + (n_pad / 32) * n_pad + 2 * n_pad * n_pad);
    return QREC_OK;
}

int qrec_info_nce_loss_grad(const float *d_S1, const float *d_S2, float div, const int32_t *d_rows, int32_t n,
                            int32_t ld, float tau, float cl_rate, void *d_workspace, float *d_out, float *d_out2,
                            double *d_loss, void *stream) {
    QREC_REQUIRE(d_S1 && d_S2 && d_workspace && d_out && d_loss && n >= 0 && div != 0.f && tau > 0.f,
                 "qrec_info_nce_loss_grad: bad argument");
    QREC_REQUIRE(n == 0 || d_rows, "qrec_info_nce_loss_grad: null row list");
    QREC_REQUIRE(n <= 16384, "qrec_info_nce_loss_grad: at most 16384 unique rows per call");
    if (n == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    float *ws = static_cast<float *>(d_workspace);
    switch (ld) {
        case 32: return run_info_nce<8>(d_S1, d_S2, div, d_rows, n, ld, tau, cl_rate, ws, d_out, d_out2, d_loss, st);
        case 64: return run_info_nce<16>(d_S1, d_S2, div, d_rows, n, ld, tau, cl_rate, ws, d_out, d_out2, d_loss, st);
        case 128: return run_info_nce<32>(d_S1, d_S2, div, d_rows, n, ld, tau, cl_rate, ws, d_out, d_out2, d_loss, st);
        case 256: return run_info_nce<64>(d_S1, d_S2, div, d_rows, n, ld, tau, cl_rate, ws, d_out, d_out2, d_loss, st);
        default: set_error("qrec_info_nce_loss_grad: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
}

int qrec_sept_ssl_workspace_bytes(int32_t n, int32_t ld, int32_t k, int64_t *bytes) {
    QREC_REQUIRE(bytes && n >= 0 && ld > 0 && k >= 1, "qrec_sept_ssl_workspace_bytes: bad argument");
    *bytes = 4 * sept_ws_floats(n, ld, k);
    return QREC_OK;
}

int qrec_sept_ssl_loss_grad(const float *d_S_friend, const float *d_S_sharing, const float *d_S_pref, const float *d_S_aug,
                            const int32_t *d_rows, int32_t n, int32_t ld, int32_t k, float ss_rate, void *d_workspace,
                            float *d_dS_friend, float *d_dS_sharing, float *d_dS_pref, float *d_dS_aug, double *d_loss,
                            int32_t *d_labels, void *d_ordered_ws, int64_t ordered_ws_bytes, void *stream) {
    QREC_REQUIRE(d_S_friend && d_S_sharing && d_S_pref && d_S_aug && d_workspace && d_dS_friend && d_dS_sharing && d_dS_pref &&
                 d_dS_aug && d_loss && n >= 0, "qrec_sept_ssl_loss_grad: bad argument");
    QREC_REQUIRE(n == 0 || d_rows, "qrec_sept_ssl_loss_grad: null row list");
    QREC_REQUIRE(n <= 16384, "qrec_sept_ssl_loss_grad: at most 16384 unique rows per call");
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(k >= 1 && k <= n, "qrec_sept_ssl_loss_grad: need 1 <= ins_cnt <= unique users in the batch (got %d, %d)", k, n);
    hipStream_t st = as_stream(stream);
    float *ws = static_cast<float *>(d_workspace);
    const float *const S[4] = {d_S_friend, d_S_sharing, d_S_pref, d_S_aug};
    float *const dS[4] = {d_dS_friend, d_dS_sharing, d_dS_pref, d_dS_aug};
    OrderedScatterWs ow = {};
    if (d_ordered_ws) {
        QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_sept_ssl_loss_grad: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
        const int rc = ordered_ws_carve(d_ordered_ws, ordered_ws_bytes, (int64_t)n * k, ld, &ow);
        if (rc != QREC_OK) return rc;
    }
    const OrderedScatterWs *owp = d_ordered_ws ? &ow : nullptr;
    switch (ld) {
        case 32: return run_sept_ssl<8>(S, d_rows, n, ld, k, ss_rate, ws, dS, d_loss, d_labels, owp, st);
        case 64: return run_sept_ssl<16>(S, d_rows, n, ld, k, ss_rate, ws, dS, d_loss, d_labels, owp, st);
        case 128: return run_sept_ssl<32>(S, d_rows, n, ld, k, ss_rate, ws, dS, d_loss, d_labels, owp, st);
        case 256: return run_sept_ssl<64>(S, d_rows, n, ld, k, ss_rate, ws, dS, d_loss, d_labels, owp, st);
        default: set_error("qrec_sept_ssl_loss_grad: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
}

}  // extern "C"
