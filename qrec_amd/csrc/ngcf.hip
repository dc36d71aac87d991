// NGCF's dense propagation layers (model/ranking/NGCF.py:27-42) and their backward pass.
//
//   side = A E (qrec_spmm_csr)
//   pre  = (side + E) W1 + (E * side) W2                       dense_fwd_kernel      (f32 MFMA)
//   act  = leaky_relu(pre, 0.2); nxt = dropout(act, keep); z = l2_normalize(nxt)
//                                                              activate_rows_kernel
//   backward:
//   dnxt = dE_next + normalize_bwd(dz);  dpre = dnxt * gate     dpre_rows_kernel
//   dA1 = dpre W1^T, dA2 = dpre W2^T;  dside = dA1 + dA2*E;  dE = dA1 + dA2*side
//                                                              dense_bwd_kernel      (f32 MFMA)
//   dW1 = (side+E)^T dpre, dW2 = (E*side)^T dpre               wgrad_kernel + wgrad_reduce_kernel (MFMA)
//   dE += A^T dside (qrec_spmm_csr with addend)
//
// Tables are [rows][ld] fp32 with ld in {32, 64, 128}; the d x d weights are stored zero-padded as
// [ld][ld].  The N x d x d products are the only GEMM-shaped work (2 N d^2 FLOP each, K = d small):
// one wavefront per 32 rows holds its A fragments in registers and sweeps the output column tiles.
#include "common.h"

using namespace qrec;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
__device__ __forceinline__ int cd_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// pre[32 rows][ld] = (side + E) W1 + (E*side) W2      NT = ld/32 output column tiles
// Persistent blocks: the two weight matrices are staged in LDS once per block (2 x LD*LD floats) and every
// wavefront walks 32-row tiles.  A operand: the lane's own row, 8 float4 loads per table and 64-column chunk;
// B operand: ds_read_b32 of W[k][32t + r] (consecutive lanes, consecutive banks).  The first version read every B
// value from global memory right before its MFMA (the compiler keeps such loads one MFMA ahead): ~400 clocks of
// load latency per 64-clock MFMA, 51 us for 1.1 GFLOP.
template <int NT>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float *__restrict__ E, const float *__restrict__ side,
                                                        const float *__restrict__ W1, const float *__restrict__ W2,
                                                        int64_t n_rows, float *__restrict__ pre) {
    constexpr int LD = 32 * NT;
    extern __shared__ float s_w[];                  // [2][LD][LD]
    for (int k = threadIdx.x; k < LD * LD / 4; k += blockDim.x) {
        reinterpret_cast<f32x4 *>(s_w)[k] = reinterpret_cast<const f32x4 *>(W1)[k];
        reinterpret_cast<f32x4 *>(s_w + LD * LD)[k] = reinterpret_cast<const f32x4 *>(W2)[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 32, row = row0 + r;
        const int64_t rowc = row < n_rows ? row : n_rows - 1;      // rows past the end are computed on a copy, not stored
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
        // ld = 128 (NT = 4) has two 64-column chunks: kept a LOOP there -- unrolled, both chunks' 2 x 32 A values are live next to the
        // 64 accumulator registers and the kernel spilled 126 VGPRs (508 B of scratch per lane)
#pragma unroll 1
        for (int c = 0; c < LD; c += 64) {
            const int k0 = c + 32 * h;
            const bool kv = k0 < LD;                 // ld = 32: the upper k-slot has no columns and feeds zeros
            const int kb = kv ? k0 : 0;
            const float keep = kv ? 1.f : 0.f;
            const f32x4 *pe = reinterpret_cast<const f32x4 *>(E + rowc * LD + kb);
            const f32x4 *ps = reinterpret_cast<const f32x4 *>(side + rowc * LD + kb);
            float a1[32], a2[32];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x4 e = pe[q], sd = ps[q];
                a1[4 * q + 0] = (sd.x + e.x) * keep; a2[4 * q + 0] = (e.x * sd.x) * keep;
                a1[4 * q + 1] = (sd.y + e.y) * keep; a2[4 * q + 1] = (e.y * sd.y) * keep;
                a1[4 * q + 2] = (sd.z + e.z) * keep; a2[4 * q + 2] = (e.z * sd.z) * keep;
                a1[4 * q + 3] = (sd.w + e.w) * keep; a2[4 * q + 3] = (e.w * sd.w) * keep;
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float *w1 = s_w + kb * LD + 32 * t + r, *w2 = w1 + LD * LD;
#pragma unroll
                for (int s = 0; s < 32; s++) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], w1[s * LD], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], w2[s * LD], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int64_t orow = row0 + cd_row(q, h);
                if (orow < n_rows) pre[orow * LD + 32 * t + r] = acc[t][q];
            }
    }
}

// One group of LPR lanes per row (float4 per lane).  In place on `pre_gate`: reads pre, writes the
// backward gate = (mask/keep) * (pre > 0 ? 1 : 0.2).  Writes nxt, z = l2_normalize(nxt) into the wide
// output table at column offset `out_off` (row stride out_ld), and 1/|nxt|.
template <int LPR>
__global__ __launch_bounds__(256) void activate_rows_kernel(float *__restrict__ pre_gate, int64_t n_rows, int d, float keep,
                                                            const float *__restrict__ mask, uint64_t seed, uint64_t stream_id,
                                                            float *__restrict__ nxt, float *__restrict__ out, int out_ld,
                                                            int out_off, float *__restrict__ inv_norm,
                                                            const int32_t *__restrict__ row_ids, const int32_t *__restrict__ n_ids,
                                                            int64_t philox_row0) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    const int64_t n_groups = (int64_t)gridDim.x * 4 * GPW;
    const int64_t n_todo = row_ids ? *n_ids : n_rows;
    for (int64_t v = gid; v < n_todo; v += n_groups) {
        const int64_t row = row_ids ? (int64_t)row_ids[v] : v;
        const int64_t off = row * (4 * LPR) + 4 * r;
        const f32x4 p = *reinterpret_cast<const f32x4 *>(pre_gate + off);
        f32x4 fac = {1.f, 1.f, 1.f, 1.f};
        if (keep < 1.f) {
            f32x4 u;
            if (mask) {
                u = *reinterpret_cast<const f32x4 *>(mask + off);       // injected 0/1 keep decisions
                fac = u / keep;
            } else {
                const int64_t grow = row + philox_row0;      // the table row this block row stands for (row-partitioned tables)
                uint32_t c[4] = {(uint32_t)grow, (uint32_t)(grow >> 32) ^ ((uint32_t)r << 8), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
                philox10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
                // tf.nn.dropout: keep iff uniform >= 1 - keep_prob
                fac.x = ((float)(c[0] >> 8) * 0x1p-24f >= 1.f - keep) ? 1.f / keep : 0.f;
                fac.y = ((float)(c[1] >> 8) * 0x1p-24f >= 1.f - keep) ? 1.f / keep : 0.f;
                fac.z = ((float)(c[2] >> 8) * 0x1p-24f >= 1.f - keep) ? 1.f / keep : 0.f;
                fac.w = ((float)(c[3] >> 8) * 0x1p-24f >= 1.f - keep) ? 1.f / keep : 0.f;
            }
        }
        auto lrelu = [](float x) { return fmaxf(0.2f * x, x); };
        f32x4 y = {lrelu(p.x) * fac.x, lrelu(p.y) * fac.y, lrelu(p.z) * fac.z, lrelu(p.w) * fac.w};
        f32x4 gate = {fac.x * (p.x > 0.f ? 1.f : 0.2f), fac.y * (p.y > 0.f ? 1.f : 0.2f),
                      fac.z * (p.z > 0.f ? 1.f : 0.2f), fac.w * (p.w > 0.f ? 1.f : 0.2f)};
        if (4 * r + 0 >= d) { y.x = 0.f; gate.x = 0.f; }
        if (4 * r + 1 >= d) { y.y = 0.f; gate.y = 0.f; }
        if (4 * r + 2 >= d) { y.z = 0.f; gate.z = 0.f; }
        if (4 * r + 3 >= d) { y.w = 0.f; gate.w = 0.f; }
        float ss = y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        ss = row_allreduce_sum<LPR>(ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        *reinterpret_cast<f32x4 *>(pre_gate + off) = gate;
        *reinterpret_cast<f32x4 *>(nxt + off) = y;
        float *o = out + row * out_ld + out_off + 4 * r;      // out_off is not 16-byte aligned in general
        if (4 * r + 0 < d) o[0] = y.x * inv;
        if (4 * r + 1 < d) o[1] = y.y * inv;
        if (4 * r + 2 < d) o[2] = y.z * inv;
        if (4 * r + 3 < d) o[3] = y.w * inv;
        if (r == 0) inv_norm[row] = inv;
    }
}

// dpre = (dE_next + (dz - z (z.dz)) * inv) * gate        (dz, z: wide tables at column offset `off`)
template <int LPR>
__global__ __launch_bounds__(256) void dpre_rows_kernel(const float *__restrict__ dE_next, const float *__restrict__ dAll,
                                                        const float *__restrict__ All, int wide_ld, int col_off,
                                                        const float *__restrict__ inv_norm, const float *__restrict__ gate,
                                                        int64_t n_rows, int d, float *__restrict__ dpre,
                                                        const int32_t *__restrict__ row_ids, const int32_t *__restrict__ n_ids,
                                                        const uint32_t *__restrict__ wide_row_mask) {
    constexpr int GPW = kWave / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    const int64_t gid = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g;
    const int64_t n_groups = (int64_t)gridDim.x * 4 * GPW;
    const int64_t n_todo = row_ids ? *n_ids : n_rows;
    for (int64_t v = gid; v < n_todo; v += n_groups) {
        const int64_t row = row_ids ? (int64_t)row_ids[v] : v;
        const int64_t off = row * (4 * LPR) + 4 * r;
        const float *dzp = dAll + row * wide_ld + col_off + 4 * r, *zp = All + row * wide_ld + col_off + 4 * r;
        f32x4 dz = {0.f, 0.f, 0.f, 0.f}, z = dz;
        // wide_row_mask: the wide gradient is only defined (and only non-zero) at the marked rows; elsewhere dz = 0
        const bool has = !wide_row_mask || ((wide_row_mask[row >> 5] >> (row & 31)) & 1u);
        if (has) {
            if (4 * r + 0 < d) { dz.x = dzp[0]; z.x = zp[0]; }
            if (4 * r + 1 < d) { dz.y = dzp[1]; z.y = zp[1]; }
            if (4 * r + 2 < d) { dz.z = dzp[2]; z.z = zp[2]; }
            if (4 * r + 3 < d) { dz.w = dzp[3]; z.w = zp[3]; }
        }
        float dot = z.x * dz.x + z.y * dz.y + z.z * dz.z + z.w * dz.w;
        dot = row_allreduce_sum<LPR>(dot);
        f32x4 dn = (dz - z * dot) * inv_norm[row];
        if (dE_next) dn = dn + *reinterpret_cast<const f32x4 *>(dE_next + off);
        const f32x4 gt = *reinterpret_cast<const f32x4 *>(gate + off);
        *reinterpret_cast<f32x4 *>(dpre + off) = dn * gt;
    }
}

// dA1 = dpre W1^T, dA2 = dpre W2^T;  dside = dA1 + dA2*E ;  dE = dA1 + dA2*side
// Same structure as dense_fwd_kernel; B[k][j] = W[j][k], so the weights are TRANSPOSED on their way into LDS
// (s_wt[k][j]) and the B reads are again consecutive lanes on consecutive banks.
template <int NT>
__global__ __launch_bounds__(256) void dense_bwd_kernel(const float *__restrict__ dpre, const float *__restrict__ W1,
                                                        const float *__restrict__ W2, const float *__restrict__ E,
                                                        const float *__restrict__ side, int64_t n_rows,
                                                        float *__restrict__ dside, float *__restrict__ dE) {
    constexpr int LD = 32 * NT, LDP = LD + 1;       // padded rows: the transposing writes below fall on distinct banks
    extern __shared__ float s_w[];                  // [2][LD][LDP], transposed
    for (int k = threadIdx.x; k < LD * LD; k += blockDim.x) {
        const int j = k / LD, c = k % LD;           // coalesced read of W[j][c], write to s_wt[c][j]
        s_w[c * LDP + j] = W1[k];
        s_w[LD * LDP + c * LDP + j] = W2[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 32, row = row0 + r;
        const int64_t rowc = row < n_rows ? row : n_rows - 1;
        f32x16 a1[NT], a2[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) { a1[t][q] = 0.f; a2[t][q] = 0.f; }
        // the epilogue's E / side values (C layout) are fetched BEFORE the MFMA loop: a wavefront owns a single tile, so
        // nothing else can hide their latency (62 -> 3x us per call when they were loaded after the products)
        float ev[NT][16], sv[NT][16];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                int64_t orow = row0 + cd_row(q, h);
                if (orow >= n_rows) orow = n_rows - 1;
                const int64_t o = orow * LD + 32 * t + r;
                ev[t][q] = E[o]; sv[t][q] = side[o];
            }
#pragma unroll
        for (int c = 0; c < LD; c += 64) {
            const int k0 = c + 32 * h;
            const bool kv = k0 < LD;
            const int kb = kv ? k0 : 0;
            const float keep = kv ? 1.f : 0.f;
            const f32x4 *pg = reinterpret_cast<const f32x4 *>(dpre + rowc * LD + kb);
            float g[32];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x4 v = pg[q];
                g[4 * q] = v.x * keep; g[4 * q + 1] = v.y * keep; g[4 * q + 2] = v.z * keep; g[4 * q + 3] = v.w * keep;
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float *w1 = s_w + kb * LDP + 32 * t + r, *w2 = w1 + LD * LDP;
#pragma unroll
                for (int s = 0; s < 32; s++) {
                    a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s], w1[s * LDP], a1[t], 0, 0, 0);
                    a2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s], w2[s * LDP], a2[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int64_t orow = row0 + cd_row(q, h);
                if (orow < n_rows) {
                    const int64_t o = orow * LD + 32 * t + r;
                    dside[o] = a1[t][q] + a2[t][q] * ev[t][q];
                    dE[o] = a1[t][q] + a2[t][q] * sv[t][q];
                }
            }
    }
}

// =============================================================================================
// LD <= 64 (NT = 1, 2): the versions that run.  Same products in the same order as dense_fwd_kernel / dense_bwd_kernel
// above (which stay for ld = 128), but
//  * the A operand no longer comes from per-lane-row global loads: a 32 x LD tile is fetched with LD/8 fully coalesced
//    float4 loads per lane (4 whole rows per instruction), parked in a wave-private LDS tile (row stride LD + 4 floats:
//    the per-row 16-byte reads of the fragment fall on distinct bank groups) and read back in fragment order; the next
//    tile's global loads are issued before the current tile's MFMA loop;
//  * the weights never change inside a launch and a wavefront's B fragments of BOTH matrices are only 2 * NT * 32
//    values per lane, so they live in registers for the whole launch (one wavefront per SIMD, a persistent loop over
//    tiles): the MFMA loop reads nothing from memory.  (B values read from LDS right before their MFMA -- 64 ds_read2
//    per tile, the round-1 kernel and the first LDS version -- cost 14 us of the kernel's 25.)
// Timeline of dense_fwd at the Yelp2018 shape (wall_clock64 stamps per wavefront, 1024 wavefronts, 2179 tiles): weights
// + first tile 3.6 us, then per tile 0.6 (park / fragment) + 4.5 (128 MFMAs) + 0.9 (stores); every wavefront has two
// tiles and 131 have a third: 21.4 us, of which 5 are that ragged third round.  Two wavefronts per SIMD without the
// prefetch: 24.6 us.
// =============================================================================================
constexpr int kTilePad = 4;

template <int LD>
struct RowTile {                                  // a wavefront's view of one 32 x LD tile
    static constexpr int RS = LD + kTilePad;      // LDS row stride (floats)
    static constexpr int LPRW = LD / 4;           // lanes per row in the load layout
    static constexpr int RPI = kWave / LPRW;      // rows per load instruction
    static constexpr int NV = 32 / RPI;           // float4 per lane per tile
    int lrow, lcol;
    __device__ explicit RowTile(int lane) : lrow(lane / LPRW), lcol(4 * (lane % LPRW)) {}
    // row_ids (may be null): the tile's rows are row_ids[row0 ...] instead of row0 ... (a listed subset of the table)
    __device__ void load(const float *__restrict__ X, int64_t row0, int64_t n_rows, f32x4 (&v)[NV],
                         const int32_t *__restrict__ row_ids = nullptr) const {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            int64_t row = row0 + k * RPI + lrow;
            if (row >= n_rows) row = n_rows - 1;                 // rows past the end: a copy of the last row, never stored
            if (row_ids) row = row_ids[row];
            v[k] = *reinterpret_cast<const f32x4 *>(X + row * LD + lcol);
        }
    }
    __device__ void park(float *tile, const f32x4 (&v)[NV]) const {
#pragma unroll
        for (int k = 0; k < NV; k++) *reinterpret_cast<f32x4 *>(tile + (k * RPI + lrow) * RS + lcol) = v[k];
    }
    // MFMA A fragment of lane (r, h): columns [32h, 32h + 32) of row r (LD = 32: the upper k-slot feeds zeros)
    __device__ static void fragment(const float *tile, int r, int h, float (&a)[32]) {
        const bool kv = 32 * h < LD;
        const float keep = kv ? 1.f : 0.f;
        const float *p = tile + r * RS + (kv ? 32 * h : 0);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * q);
            a[4 * q] = v.x * keep; a[4 * q + 1] = v.y * keep; a[4 * q + 2] = v.z * keep; a[4 * q + 3] = v.w * keep;
        }
    }
};

template <int NT>
__global__ __launch_bounds__(256) void dense_fwd_lds_kernel(const float *__restrict__ E, const float *__restrict__ side,
                                                            const float *__restrict__ W1, const float *__restrict__ W2,
                                                            int64_t n_all, float *__restrict__ pre,
                                                            const int32_t *__restrict__ row_ids, const int32_t *__restrict__ n_ids) {
    constexpr int LD = 32 * NT;
    using Tile = RowTile<LD>;
    extern __shared__ float s_mem[];                // one tile per wavefront
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    float *tile_mem = s_mem + (threadIdx.x >> 6) * (32 * Tile::RS);
    const Tile tl(lane);
    const int kb = 32 * h < LD ? 32 * h : 0;        // ld = 32: the upper k-slot has no columns; its A values are zeros
    const int64_t n_rows = row_ids ? *n_ids : n_all;            // rows to process: the listed ones, or all
    const int64_t n_tiles = (n_rows + 31) / 32, stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 e[Tile::NV], sd[Tile::NV];
    if (tile < n_tiles) { tl.load(E, tile * 32, n_rows, e, row_ids); tl.load(side, tile * 32, n_rows, sd, row_ids); }
    float b1[NT][32], b2[NT][32];                   // B fragments: W[kb + s][32 t + r]
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int s = 0; s < 32; s++) {
            b1[t][s] = W1[(kb + s) * LD + 32 * t + r];
            b2[t][s] = W2[(kb + s) * LD + 32 * t + r];
        }
    for (; tile < n_tiles; tile += stride) {
        const int64_t row0 = tile * 32;
        f32x4 t1[Tile::NV], t2[Tile::NV];
#pragma unroll
        for (int k = 0; k < Tile::NV; k++) { t1[k] = sd[k] + e[k]; t2[k] = e[k] * sd[k]; }
        float a1[32], a2[32];
        tl.park(tile_mem, t1); Tile::fragment(tile_mem, r, h, a1);
        tl.park(tile_mem, t2); Tile::fragment(tile_mem, r, h, a2);     // same wavefront, LDS operations complete in order
        if (tile + stride < n_tiles) { tl.load(E, (tile + stride) * 32, n_rows, e, row_ids); tl.load(side, (tile + stride) * 32, n_rows, sd, row_ids); }
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int s = 0; s < 32; s++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[t][s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], b2[t][s], acc[t], 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            int64_t orow = row0 + cd_row(q, h);
            if (orow < n_rows) {
                if (row_ids) orow = row_ids[orow];
#pragma unroll
                for (int t = 0; t < NT; t++) pre[orow * LD + 32 * t + r] = acc[t][q];
            }
        }
    }
}

// B[k][j] = W[j][k]: lane (r, h) holds W[32 t + r][kb .. kb + 32) of both matrices -- its own row, 8 float4 loads each.
template <int NT>
__global__ __launch_bounds__(256) void dense_bwd_lds_kernel(const float *__restrict__ dpre, const float *__restrict__ W1,
                                                            const float *__restrict__ W2, const float *__restrict__ E,
                                                            const float *__restrict__ side, int64_t n_all,
                                                            float *__restrict__ dside, float *__restrict__ dE,
                                                            const int32_t *__restrict__ row_ids, const int32_t *__restrict__ n_ids) {
    constexpr int LD = 32 * NT;
    using Tile = RowTile<LD>;
    extern __shared__ float s_mem[];                // one tile per wavefront
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    float *tile_mem = s_mem + (threadIdx.x >> 6) * (32 * Tile::RS);
    const Tile tl(lane);
    const int kb = 32 * h < LD ? 32 * h : 0;
    const int64_t n_rows = row_ids ? *n_ids : n_all;
    const int64_t n_tiles = (n_rows + 31) / 32, stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 gn[Tile::NV];
    if (tile < n_tiles) tl.load(dpre, tile * 32, n_rows, gn, row_ids);
    float b1[NT][32], b2[NT][32];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(W1 + (32 * t + r) * LD + kb + 4 * q);
            const f32x4 v2 = *reinterpret_cast<const f32x4 *>(W2 + (32 * t + r) * LD + kb + 4 * q);
            b1[t][4 * q] = v1.x; b1[t][4 * q + 1] = v1.y; b1[t][4 * q + 2] = v1.z; b1[t][4 * q + 3] = v1.w;
            b2[t][4 * q] = v2.x; b2[t][4 * q + 1] = v2.y; b2[t][4 * q + 2] = v2.z; b2[t][4 * q + 3] = v2.w;
        }
    for (; tile < n_tiles; tile += stride) {
        const int64_t row0 = tile * 32;
        float g[32];
        tl.park(tile_mem, gn); Tile::fragment(tile_mem, r, h, g);
        // the epilogue's E / side values (C layout: 128-byte runs) and the next tile go out before the MFMA loop
        float ev[NT][16], sv[NT][16];
        int64_t prow[16];                              // physical rows of this lane's 16 output rows
#pragma unroll
        for (int q = 0; q < 16; q++) {
            int64_t orow = row0 + cd_row(q, h);
            if (orow >= n_rows) orow = n_rows - 1;
            prow[q] = row_ids ? (int64_t)row_ids[orow] : orow;
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int64_t o = prow[q] * LD + 32 * t + r;
                ev[t][q] = E[o]; sv[t][q] = side[o];
            }
        if (tile + stride < n_tiles) tl.load(dpre, (tile + stride) * 32, n_rows, gn, row_ids);
        f32x16 a1[NT], a2[NT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) { a1[t][q] = 0.f; a2[t][q] = 0.f; }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int s = 0; s < 32; s++) {
                a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s], b1[t][s], a1[t], 0, 0, 0);
                a2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[s], b2[t][s], a2[t], 0, 0, 0);
            }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (row0 + cd_row(q, h) < n_rows) {
                    const int64_t o = prow[q] * LD + 32 * t + r;
                    dside[o] = a1[t][q] + a2[t][q] * ev[t][q];
                    dE[o] = a1[t][q] + a2[t][q] * sv[t][q];
                }
            }
    }
}

// Weight gradients, LD <= 64.  A block owns kBlockRows rows; its 256 threads fetch a stage of 32 rows of E, side and
// dpre with coalesced float4 loads (6 per thread) into a double-buffered LDS stage while the previous stage is being
// multiplied; wavefront w owns the output block (which = w / NT, ti = w % NT) -- 32 rows of gW_which, all LD columns --
// so nothing is summed across wavefronts.  Per stage and wavefront: 16 k-steps (row 2s + h of the stage), two LDS reads
// for the A value ((side + E) or E * side at column 32 ti + r), NT for B, NT MFMAs.
// partial[slab][which][i][j] as before (slab = block), then wgrad_reduce_kernel.
constexpr int kBlockRows = 128;
template <int NT>
__global__ __launch_bounds__(256) void wgrad_lds_kernel(const float *__restrict__ E, const float *__restrict__ side,
                                                        const float *__restrict__ dpre, int64_t n_all,
                                                        float *__restrict__ partial, const int32_t *__restrict__ row_ids,
                                                        const int32_t *__restrict__ n_ids) {
    constexpr int LD = 32 * NT, RS = LD + kTilePad;
    const int64_t n_rows = row_ids ? *n_ids : n_all;
    constexpr int NV = 32 * LD / 4 / 256;          // float4 per thread, array and stage (LD = 64: 2, LD = 32: 1)
    constexpr int kStage = 3 * 32 * RS;            // floats per stage buffer: E, side, dpre
    extern __shared__ float s_mem[];               // [2][3][32][RS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int which = wave / NT, ti = wave % NT;
    const bool owner = wave < 2 * NT;              // LD = 32: two output blocks, wavefronts 2 and 3 only help fetching
    const int64_t n0 = (int64_t)blockIdx.x * kBlockRows;
    f32x4 ve[NV], vs[NV], vd[NV];
    auto fetch = [&](int st) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int idx = threadIdx.x + 256 * k, row = idx / (LD / 4), c4 = 4 * (idx % (LD / 4));
            int64_t n = n0 + 32 * st + row;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if (n < n_rows) {
                if (row_ids) n = row_ids[n];
                ve[k] = *reinterpret_cast<const f32x4 *>(E + n * LD + c4);
                vs[k] = *reinterpret_cast<const f32x4 *>(side + n * LD + c4);
                vd[k] = *reinterpret_cast<const f32x4 *>(dpre + n * LD + c4);
            } else { ve[k] = zero; vs[k] = zero; vd[k] = zero; }      // rows past the end contribute 0
        }
    };
    auto park = [&](int buf) {
        float *b = s_mem + buf * kStage;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int idx = threadIdx.x + 256 * k, row = idx / (LD / 4), c4 = 4 * (idx % (LD / 4));
            *reinterpret_cast<f32x4 *>(b + row * RS + c4) = ve[k];
            *reinterpret_cast<f32x4 *>(b + 32 * RS + row * RS + c4) = vs[k];
            *reinterpret_cast<f32x4 *>(b + 64 * RS + row * RS + c4) = vd[k];
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    constexpr int kStages = kBlockRows / 32;
    fetch(0); park(0);
    __syncthreads();
    for (int st = 0; st < kStages; st++) {
        const bool more = st + 1 < kStages && n0 + 32 * (st + 1) < n_rows;
        if (more) fetch(st + 1);
        if (owner) {
            const float *b = s_mem + (st & 1) * kStage;
            const float *pe = b + h * RS + 32 * ti + r, *ps = pe + 32 * RS, *pd = b + 64 * RS + h * RS + r;
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float e = pe[2 * s * RS], sd = ps[2 * s * RS];
                const float a = which == 0 ? sd + e : e * sd;
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pd[2 * s * RS + 32 * t], acc[t], 0, 0, 0);
            }
        }
        if (more) park((st + 1) & 1);
        __syncthreads();
        if (!more) break;
    }
    if (owner) {
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++)
                partial[(((int64_t)blockIdx.x * 2 + which) * LD + 32 * ti + cd_row(q, h)) * LD + 32 * t + r] = acc[t][q];
    }
}

// partial[slab][which][i][j] = sum over the slab's rows n of A_which[n][i] * dpre[n][j]
//   which = 0: A = side + E ; 1: A = E * side.
// One wavefront per (slab of 128 rows, 32-column block ti of A): it forms BOTH products for ALL NT column tiles of
// dpre, i.e. 2*NT MFMAs per k-step from 2 + NT coalesced dword loads (the first version issued 3 loads per MFMA from
// one wavefront per SIMD and sat in load latency: 98 us for 1.1 GFLOP).  A chunk's 32 k-steps are fetched first, then
// its MFMAs run.  Rows past the end feed a = 0.
constexpr int kWaveRows = 128;                 // rows one wavefront accumulates (64 gives the same 36 us and doubles the partial slabs)
constexpr int kSlabRows = 4 * kWaveRows;       // rows per block = per partial slab (4 wavefronts, summed through LDS)
template <int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const float *__restrict__ E, const float *__restrict__ side,
                                                    const float *__restrict__ dpre, int64_t n_rows,
                                                    float *__restrict__ partial) {
    constexpr int LD = 32 * NT;
    extern __shared__ float s_acc[];               // [4 waves][2*NT tiles][16][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int slab = blockIdx.x, ti = blockIdx.y;
    const int64_t n0 = (int64_t)slab * kSlabRows + (int64_t)wave * kWaveRows;
    f32x16 acc[2][NT];
#pragma unroll
    for (int w = 0; w < 2; w++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[w][t][q] = 0.f;
    const float *pe = E + 32 * ti + r, *ps = side + 32 * ti + r, *pd = dpre + r;
    for (int c = 0; c < kWaveRows; c += 64) {
        float a0[32], a1[32], b[NT][32];
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const int64_t n = n0 + c + 32 * h + s;
            const int64_t nc = n < n_rows ? n : n_rows - 1;
            const float keep = n < n_rows ? 1.f : 0.f;
            const float e = pe[nc * LD], sd = ps[nc * LD];
            a0[s] = (sd + e) * keep; a1[s] = (e * sd) * keep;
#pragma unroll
            for (int t = 0; t < NT; t++) b[t][s] = pd[nc * LD + 32 * t];
        }
#pragma unroll
        for (int s = 0; s < 32; s++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b[t][s], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b[t][s], acc[1][t], 0, 0, 0);
            }
    }
    // the block's four wavefronts hold the same tiles over different rows: sum them in wave order (deterministic)
#pragma unroll
    for (int w = 0; w < 2; w++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) s_acc[((wave * 2 * NT + w * NT + t) * 16 + q) * 64 + lane] = acc[w][t][q];
    __syncthreads();
    constexpr int kPerWave = 2 * NT * 16 * 64;     // floats one wavefront deposited
    for (int k = threadIdx.x; k < kPerWave; k += 256) {
        const float v = ((s_acc[k] + s_acc[kPerWave + k]) + s_acc[2 * kPerWave + k]) + s_acc[3 * kPerWave + k];
        const int ln = k & 63, q = (k >> 6) & 15, wt = k >> 10, w = wt / NT, t = wt % NT;
        partial[(((int64_t)slab * 2 + w) * LD + 32 * ti + cd_row(q, ln >> 5)) * LD + 32 * t + (ln & 31)] = v;
    }
}

// gW[which][i][j] = sum_slab partial[slab][which][i][j].  A thread owns four consecutive elements (float4) of one of
// kSplit slab classes (slabs c, c + kSplit, ...: coalesced 16-byte loads, added in slab order); the kSplit class sums
// of an element meet in LDS and are added in class order -- deterministic.
constexpr int kSplit = 16;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int n_slabs, int ld,
                                                           float *__restrict__ gW1, float *__restrict__ gW2) {
    __shared__ f32x4 s_part[256];
    const int n4 = 2 * ld * ld / 4;                                     // float4 elements of [2][ld][ld]
    const int e4 = blockIdx.x * (256 / kSplit) + threadIdx.x / kSplit, c = threadIdx.x % kSplit;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e4 < n4)
        for (int s = c; s < n_slabs; s += kSplit) acc = acc + reinterpret_cast<const f32x4 *>(partial)[(int64_t)s * n4 + e4];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (c == 0 && e4 < n4) {
        f32x4 t = s_part[threadIdx.x];
#pragma unroll
        for (int k = 1; k < kSplit; k++) t = t + s_part[threadIdx.x + k];
        const int per = ld * ld / 4;
        reinterpret_cast<f32x4 *>(e4 < per ? gW1 : gW2)[e4 < per ? e4 : e4 - per] = t;
    }
}

// dst[row][c] (=|+=) src[row][off + c], c < d     (the ego block of the wide table, in and out).  row_ids (may be null):
// only the listed rows are touched
__global__ __launch_bounds__(256) void add_cols_kernel(float *__restrict__ dst, int dst_ld, const float *__restrict__ src,
                                                       int src_ld, int off, int64_t n_rows, int d, int assign,
                                                       const int32_t *__restrict__ row_ids, const int32_t *__restrict__ n_ids) {
    const int64_t total = (row_ids ? (int64_t)*n_ids : n_rows) * d;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row_ids ? (int64_t)row_ids[k / d] : k / d; const int c = (int)(k % d);
        const float v = src[row * src_ld + off + c];
        if (assign) dst[row * dst_ld + c] = v; else dst[row * dst_ld + c] += v;
    }
}

// X[row][0 .. ld) = 0 for the listed rows: one float4 per thread
__global__ __launch_bounds__(256) void zero_rows_kernel(float *__restrict__ X, int ld, const int32_t *__restrict__ row_ids,
                                                        const int32_t *__restrict__ n_ids) {
    const int per_row = ld / 4;
    const int64_t total = (int64_t)*n_ids * per_row;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x)
        *reinterpret_cast<f32x4 *>(X + (int64_t)row_ids[k / per_row] * ld + 4 * (k % per_row)) = zero;
}

// rows[0 .. count) = the set bits of a row bitmap, ascending.  One block: thread t owns a contiguous run of mask words,
// counts its bits, the block scans the counts, every thread writes its rows.
__global__ __launch_bounds__(1024) void compact_rows_kernel(const uint32_t *__restrict__ mask, int64_t n_rows,
                                                            int32_t *__restrict__ rows, int32_t *__restrict__ count, int capacity) {
    __shared__ int s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_words = (n_rows + 31) / 32;
    const int64_t per = (n_words + 1023) / 1024, w0 = per * threadIdx.x, w1 = w0 + per < n_words ? w0 + per : n_words;
    auto word = [&](int64_t w) {
        uint32_t m = mask[w];
        if (w == n_words - 1 && (n_rows & 31)) m &= (1u << (n_rows & 31)) - 1u;      // bits past the last row do not count
        return m;
    };
    int mine = 0;
    for (int64_t w = w0; w < w1; w++) mine += __popc(word(w));
    int incl = mine;                                     // inclusive scan inside the wavefront, then over the 16 wavefronts
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, kWave); if (lane >= off) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int v = s_wave[k]; if (k < wave) base += v; total += v; }
    int pos = base + incl - mine;
    for (int64_t w = w0; w < w1; w++) {
        uint32_t m = word(w);
        while (m) {
            const int b = __ffs(m) - 1; m &= m - 1;
            if (pos < capacity) rows[pos] = (int32_t)(w * 32 + b);
            pos++;
        }
    }
    if (threadIdx.x == 0) *count = total < capacity ? total : capacity;
}

// The whole "which rows does this batch touch" step in one launch, for tables whose bitmap fits the LDS: clear the
// bitmap, mark u / n_users + i / n_users + j, publish the bitmap, emit the ascending row list.  (As three launches --
// memset, qrec_mark_batch_rows, compaction -- it is three launch latencies, ~14 us, for a few KB of work.)
__global__ __launch_bounds__(1024) void mark_compact_kernel(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                                            const int32_t *__restrict__ j, int B, int n_users, int64_t n_rows,
                                                            uint32_t *__restrict__ mask_out, int32_t *__restrict__ rows,
                                                            int32_t *__restrict__ count, int capacity,
                                                            double *__restrict__ zero8, int n_zero8) {
    extern __shared__ uint32_t s_mask[];               // n_words words, then 16 wave sums
    if ((int)threadIdx.x < n_zero8) zero8[threadIdx.x] = 0.0;      // the step's scalar accumulators (loss terms): no memset launch for 8 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_words = (n_rows + 31) / 32;
    int *s_wave = reinterpret_cast<int *>(s_mask + n_words);
    for (int64_t w = threadIdx.x; w < n_words; w += 1024) s_mask[w] = 0u;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 1024) {
        const int ru = u[b], ri = n_users + i[b], rj = n_users + j[b];
        atomicOr(&s_mask[ru >> 5], 1u << (ru & 31));
        atomicOr(&s_mask[ri >> 5], 1u << (ri & 31));
        atomicOr(&s_mask[rj >> 5], 1u << (rj & 31));
    }
    __syncthreads();
    for (int64_t w = threadIdx.x; w < n_words; w += 1024) mask_out[w] = s_mask[w];
    const int64_t per = (n_words + 1023) / 1024, w0 = per * threadIdx.x, w1 = w0 + per < n_words ? w0 + per : n_words;
    int mine = 0;
    for (int64_t w = w0; w < w1; w++) mine += __popc(s_mask[w]);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, kWave); if (lane >= off) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int v = s_wave[k]; if (k < wave) base += v; total += v; }
    int pos = base + incl - mine;
    for (int64_t w = w0; w < w1; w++) {
        uint32_t m = s_mask[w];
        while (m) {
            const int b = __ffs(m) - 1; m &= m - 1;
            if (rows && pos < capacity) rows[pos] = (int32_t)(w * 32 + b);
            pos++;
        }
    }
    if (count && threadIdx.x == 0) *count = total < capacity ? total : capacity;
}
constexpr int64_t kLdsMaskRows = 1 << 20;              // 128 KB of bitmap

// tf.unique of every batch of an epoch's id stream in one launch (SimGCL.py:61-64 on a device-drawn batch stream): block b
// marks ids[b * batch ...] in an LDS bitmap over [0, id_range) and emits the distinct ids, ascending and shifted by
// out_offset, to rows[b * batch ...]; counts[b] = how many.  (tf.unique keeps first-appearance order; the InfoNCE sums
// that consume the list do not depend on the order beyond fp32 rounding.)
__global__ __launch_bounds__(1024) void unique_per_batch_kernel(const int32_t *__restrict__ ids, int64_t n, int batch, int id_range,
                                                                int out_offset, int32_t *__restrict__ rows, int32_t *__restrict__ counts) {
    extern __shared__ uint32_t s_mask[];               // n_words words, then 16 wave sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_words = (id_range + 31) / 32;
    int *s_wave = reinterpret_cast<int *>(s_mask + n_words);
    const int64_t b0 = (int64_t)blockIdx.x * batch;
    const int B = (int)(n - b0 < batch ? n - b0 : batch);
    for (int w = threadIdx.x; w < n_words; w += 1024) s_mask[w] = 0u;
    __syncthreads();
    for (int k = threadIdx.x; k < B; k += 1024) { const int v = ids[b0 + k]; atomicOr(&s_mask[v >> 5], 1u << (v & 31)); }
    __syncthreads();
    const int per = (n_words + 1023) / 1024, w0 = per * (int)threadIdx.x, w1 = w0 + per < n_words ? w0 + per : n_words;
    int mine = 0;
    for (int w = w0; w < w1; w++) mine += __popc(s_mask[w]);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, kWave); if (lane >= off) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int v = s_wave[k]; if (k < wave) base += v; total += v; }
    int pos = base + incl - mine;
    for (int w = w0; w < w1; w++) {
        uint32_t m = s_mask[w];
        while (m) {
            const int b = __ffs(m) - 1; m &= m - 1;
            rows[b0 + pos++] = w * 32 + b + out_offset;
        }
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

// persistent grid of the dense-layer kernels: every block stages the weights in LDS once, so no more blocks than
// the chip keeps resident (2 per CU; 1 when the weights take 128 KB), and never more than there are 128-row groups
unsigned dense_grid(int64_t n_rows, int ld, bool one_per_cu = false) {
    const int64_t groups = (n_rows + 127) / 128;
    const int64_t resident = ld >= 128 || one_per_cu ? 256 : 512;
    return (unsigned)(groups < resident ? groups : resident);
}
hipError_t allow_big_lds(const void *kernel, size_t bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" {

int qrec_ngcf_dense_fwd(const float *d_E, const float *d_side, const float *d_W1, const float *d_W2, int64_t n_rows,
                        int32_t ld, float *d_pre, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                        void *stream) {
    QREC_REQUIRE(d_E && d_side && d_W1 && d_W2 && d_pre && n_rows >= 0, "qrec_ngcf_dense_fwd: bad argument");
    QREC_REQUIRE(!d_row_ids || (d_n_row_ids && max_row_ids >= 0 && ld <= 64), "qrec_ngcf_dense_fwd: a row subset needs its count, a bound, and ld <= 64");
    const int64_t work_rows = d_row_ids ? max_row_ids : n_rows;
    if (work_rows == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    const unsigned blocks = dense_grid(n_rows, ld);
    const size_t lds = (size_t)2 * ld * ld * sizeof(float);
    const unsigned pblocks = dense_grid(work_rows, ld, true);       // ld <= 64: weights in registers, one wavefront per SIMD
    const size_t lds_tiles = (size_t)4 * 32 * (ld + kTilePad) * sizeof(float);
    switch (ld) {
        case 32: hipLaunchKernelGGL(dense_fwd_lds_kernel<1>, dim3(pblocks), dim3(256), lds_tiles, st, d_E, d_side, d_W1, d_W2, n_rows, d_pre, d_row_ids, d_n_row_ids); break;
        case 64: hipLaunchKernelGGL(dense_fwd_lds_kernel<2>, dim3(pblocks), dim3(256), lds_tiles, st, d_E, d_side, d_W1, d_W2, n_rows, d_pre, d_row_ids, d_n_row_ids); break;
        case 128:
            QREC_HIP_CHECK(allow_big_lds(reinterpret_cast<const void *>(&dense_fwd_kernel<4>), lds));
            hipLaunchKernelGGL(dense_fwd_kernel<4>, dim3(blocks), dim3(256), lds, st, d_E, d_side, d_W1, d_W2, n_rows, d_pre); break;
        default: set_error("qrec_ngcf_dense_fwd: row stride must be 32, 64 or 128 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_ngcf_activate(float *d_pre_gate, int64_t n_rows, int32_t d, int32_t ld, float keep, const float *d_mask,
                       uint64_t seed, uint64_t stream_id, float *d_next, float *d_wide, int32_t wide_ld,
                       int32_t col_off, float *d_inv_norm, const int32_t *d_row_ids, const int32_t *d_n_row_ids,
                       int32_t max_row_ids, int64_t philox_row0, void *stream) {
    QREC_REQUIRE(d_pre_gate && d_next && d_wide && d_inv_norm && n_rows >= 0 && d >= 1 && ld >= d && keep > 0.f && keep <= 1.f,
                 "qrec_ngcf_activate: bad argument");
    QREC_REQUIRE(col_off >= 0 && col_off + d <= wide_ld, "qrec_ngcf_activate: column block outside the wide table");
    QREC_REQUIRE(!d_row_ids || (d_n_row_ids && max_row_ids >= 0), "qrec_ngcf_activate: a row subset needs its count and a bound");
    const int64_t work_rows = d_row_ids ? max_row_ids : n_rows;
    if (work_rows == 0) return QREC_OK;
    hipStream_t st = as_stream(stream);
    int64_t blocks;
#define QREC_ACT(LPR)                                                                                              \
    blocks = (work_rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 2048) blocks = 2048;                  \
    hipLaunchKernelGGL((activate_rows_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, d_pre_gate, n_rows, d, keep, \
                       d_mask, seed, stream_id, d_next, d_wide, wide_ld, col_off, d_inv_norm, d_row_ids, d_n_row_ids, philox_row0)
    switch (ld) {
        case 32: QREC_ACT(8); break;
        case 64: QREC_ACT(16); break;
        case 128: QREC_ACT(32); break;
        default: set_error("qrec_ngcf_activate: row stride must be 32, 64 or 128 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_ACT
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_ngcf_layer_bwd(const float *d_dE_next, const float *d_dWide, const float *d_wide, int32_t wide_ld,
                        int32_t col_off, const float *d_inv_norm, const float *d_gate, const float *d_E,
                        const float *d_side, const float *d_W1, const float *d_W2, int64_t n_rows, int32_t d,
                        int32_t ld, float *d_dpre, float *d_dside, float *d_dE, float *d_partial, float *d_gW1,
                        float *d_gW2, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                        const uint32_t *d_wide_row_mask, void *stream) {
    QREC_REQUIRE(d_dWide && d_wide && d_inv_norm && d_gate && d_E && d_side && d_W1 && d_W2 && d_dpre && d_dside && d_dE &&
                     d_partial && d_gW1 && d_gW2 && n_rows > 0, "qrec_ngcf_layer_bwd: bad argument");
    QREC_REQUIRE(!d_row_ids || (d_n_row_ids && max_row_ids > 0 && ld <= 64), "qrec_ngcf_layer_bwd: a row subset needs its count, a bound > 0, and ld <= 64");
    const int64_t work_rows = d_row_ids ? max_row_ids : n_rows;
    hipStream_t st = as_stream(stream);
    int64_t blocks;
#define QREC_DP(LPR)                                                                                              \
    blocks = (work_rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 2048) blocks = 2048;                 \
    hipLaunchKernelGGL((dpre_rows_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, d_dE_next, d_dWide, d_wide, \
                       wide_ld, col_off, d_inv_norm, d_gate, n_rows, d, d_dpre, d_row_ids, d_n_row_ids, d_wide_row_mask)
    const unsigned gblocks = dense_grid(work_rows, ld, ld <= 64);   // ld <= 64: weights in registers, one wavefront per SIMD
    const size_t lds = (size_t)2 * ld * (ld + 1) * sizeof(float);
    const size_t lds_tiles = (size_t)4 * 32 * (ld + kTilePad) * sizeof(float);
    switch (ld) {
        case 32: QREC_DP(8); hipLaunchKernelGGL(dense_bwd_lds_kernel<1>, dim3(gblocks), dim3(256), lds_tiles, st, d_dpre, d_W1, d_W2, d_E, d_side, n_rows, d_dside, d_dE, d_row_ids, d_n_row_ids); break;
        case 64: QREC_DP(16); hipLaunchKernelGGL(dense_bwd_lds_kernel<2>, dim3(gblocks), dim3(256), lds_tiles, st, d_dpre, d_W1, d_W2, d_E, d_side, n_rows, d_dside, d_dE, d_row_ids, d_n_row_ids); break;
        case 128:
            QREC_DP(32);
            QREC_HIP_CHECK(allow_big_lds(reinterpret_cast<const void *>(&dense_bwd_kernel<4>), lds));
            hipLaunchKernelGGL(dense_bwd_kernel<4>, dim3(gblocks), dim3(256), lds, st, d_dpre, d_W1, d_W2, d_E, d_side, n_rows, d_dside, d_dE); break;
        default: set_error("qrec_ngcf_layer_bwd: row stride must be 32, 64 or 128 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_DP
    QREC_LAUNCH_CHECK();
    const int nt = ld / 32;
    const int n_slabs = (int)(nt == 4 ? (n_rows + kSlabRows - 1) / kSlabRows : (work_rows + kBlockRows - 1) / kBlockRows);
    const size_t wlds = nt == 4 ? (size_t)4 * 2 * nt * 16 * 64 * sizeof(float) : (size_t)2 * 3 * 32 * (ld + kTilePad) * sizeof(float);
    if (nt == 1) hipLaunchKernelGGL(wgrad_lds_kernel<1>, dim3((unsigned)n_slabs), dim3(256), wlds, st, d_E, d_side, d_dpre, n_rows, d_partial, d_row_ids, d_n_row_ids);
    else if (nt == 2) hipLaunchKernelGGL(wgrad_lds_kernel<2>, dim3((unsigned)n_slabs), dim3(256), wlds, st, d_E, d_side, d_dpre, n_rows, d_partial, d_row_ids, d_n_row_ids);
    else {
        QREC_HIP_CHECK(allow_big_lds(reinterpret_cast<const void *>(&wgrad_kernel<4>), wlds));
        hipLaunchKernelGGL(wgrad_kernel<4>, dim3((unsigned)n_slabs, 4), dim3(256), wlds, st, d_E, d_side, d_dpre, n_rows, d_partial);
    }
    QREC_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((2 * ld * ld / 4 + 256 / kSplit - 1) / (256 / kSplit))), dim3(256), 0, st,
                       d_partial, n_slabs, ld, d_gW1, d_gW2);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_ngcf_wgrad_partial_bytes(int64_t n_rows, int32_t ld, int64_t *bytes) {
    QREC_REQUIRE(bytes && n_rows >= 0 && ld > 0, "qrec_ngcf_wgrad_partial_bytes: bad argument");
    const int64_t slab_rows = ld >= 128 ? kSlabRows : kBlockRows;
    *bytes = ((n_rows + slab_rows - 1) / slab_rows) * 2 * (int64_t)ld * ld * 4;
    return QREC_OK;
}

int qrec_compact_marked_rows(const uint32_t *d_row_mask, int64_t n_rows, int32_t *d_rows, int32_t *d_count, int32_t capacity,
                             void *stream) {
    QREC_REQUIRE(d_row_mask && d_rows && d_count && n_rows >= 0 && n_rows < (1ll << 31) && capacity >= 0, "qrec_compact_marked_rows: bad argument");
    hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, as_stream(stream), d_row_mask, n_rows, d_rows, d_count, capacity);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_mark_compact_batch_rows(const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int32_t B, int32_t n_users,
                                 int64_t n_rows, uint32_t *d_row_mask, int32_t *d_rows, int32_t *d_count, int32_t capacity,
                                 double *d_zero8, int32_t n_zero8, void *stream) {
    QREC_REQUIRE(d_row_mask && B >= 0 && n_users >= 0 && n_rows >= n_users && n_rows < (1ll << 31) && capacity >= 0 &&
                 (!d_rows == !d_count) && n_zero8 >= 0 && n_zero8 <= 64 && (n_zero8 == 0 || d_zero8),
                 "qrec_mark_compact_batch_rows: bad argument");
    QREC_REQUIRE(B == 0 || (d_u && d_i && d_j), "qrec_mark_compact_batch_rows: null index array");
    hipStream_t st = as_stream(stream);
    const int64_t n_words = (n_rows + 31) / 32;
    if (n_rows <= kLdsMaskRows) {
        const size_t lds = (size_t)n_words * 4 + 64;
        QREC_HIP_CHECK(allow_big_lds(reinterpret_cast<const void *>(&mark_compact_kernel), lds));
        hipLaunchKernelGGL(mark_compact_kernel, dim3(1), dim3(1024), lds, st, d_u, d_i, d_j, B, n_users, n_rows, d_row_mask, d_rows,
                           d_count, capacity, d_zero8, n_zero8);
        QREC_LAUNCH_CHECK();
        return QREC_OK;
    }
    if (n_zero8) QREC_HIP_CHECK(hipMemsetAsync(d_zero8, 0, (size_t)n_zero8 * 8, st));
    QREC_HIP_CHECK(hipMemsetAsync(d_row_mask, 0, (size_t)n_words * 4, st));
    int rc = qrec_mark_batch_rows(d_u, d_i, d_j, B, n_users, d_row_mask, stream);
    if (rc != QREC_OK || !d_rows) return rc;
    return qrec_compact_marked_rows(d_row_mask, n_rows, d_rows, d_count, capacity, stream);
}

int qrec_unique_per_batch(const int32_t *d_ids, int64_t n, int32_t batch, int32_t id_range, int32_t out_offset, int32_t *d_rows,
                          int32_t *d_counts, void *stream) {
    QREC_REQUIRE(n >= 0 && batch >= 1 && id_range >= 1 && id_range <= kLdsMaskRows, "qrec_unique_per_batch: bad sizes (id_range <= 2^20)");
    QREC_REQUIRE(n == 0 || (d_ids && d_rows && d_counts), "qrec_unique_per_batch: null argument");
    if (n == 0) return QREC_OK;
    const size_t lds = (size_t)((id_range + 31) / 32) * 4 + 64;
    QREC_HIP_CHECK(allow_big_lds(reinterpret_cast<const void *>(&unique_per_batch_kernel), lds));
    hipLaunchKernelGGL(unique_per_batch_kernel, dim3((unsigned)((n + batch - 1) / batch)), dim3(1024), lds, as_stream(stream), d_ids, n, batch,
                       id_range, out_offset, d_rows, d_counts);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_copy_cols(float *d_dst, int32_t dst_ld, const float *d_src, int32_t src_ld, int32_t src_col_off, int64_t n_rows,
                   int32_t d, int32_t accumulate, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                   void *stream) {
    QREC_REQUIRE(d_dst && d_src && n_rows >= 0 && d >= 1 && d <= dst_ld && src_col_off >= 0 && src_col_off + d <= src_ld,
                 "qrec_copy_cols: bad argument");
    QREC_REQUIRE(!d_row_ids || (d_n_row_ids && max_row_ids >= 0), "qrec_copy_cols: a row subset needs its count and a bound");
    const int64_t work_rows = d_row_ids ? max_row_ids : n_rows;
    if (work_rows == 0) return QREC_OK;
    int64_t blocks = (work_rows * d + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_dst, dst_ld, d_src, src_ld,
                       src_col_off, n_rows, d, accumulate ? 0 : 1, d_row_ids, d_n_row_ids);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

int qrec_zero_rows(float *d_X, int32_t ld, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids, void *stream) {
    QREC_REQUIRE(d_X && d_row_ids && d_n_row_ids && ld > 0 && ld % 4 == 0 && max_row_ids >= 0, "qrec_zero_rows: bad argument");
    if (max_row_ids == 0) return QREC_OK;
    int64_t blocks = ((int64_t)max_row_ids * (ld / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_X, ld, d_row_ids, d_n_row_ids);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // extern "C"
