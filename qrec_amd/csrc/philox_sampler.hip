// Throughput-mode negative sampler on the device.
//
// Draws, for every training triplet t = (u, i), one item j uniformly from the items that
// are NOT positives of u -- the distribution of the reference's rejection loop
// (model/ranking/BPR.py:35-37) -- with a counter-based generator so that every triplet
// is independent of every other (no serial stream to replay):
//     words = Philox4x32-10(counter = {t_lo, t_hi, block, epoch_lo}, key = {seed_lo, seed_hi ^ epoch_hi})
//     candidate = word >> (32 - bit_length(n_items)); rejected if >= n_items   (unbiased)
//     rejected if candidate in positives(u)   (binary search in the user's sorted CSR row)
// The result depends only on (seed, epoch, t): reproducible, order-free, identical on any
// number of GPUs.  It is NOT the CPython stream (use qrec_mt_bpr_sample_epoch for that).
//
// Memory: 12 B read (row_user + indptr pair is L2 resident) + ~log2(deg) 4-B probes that
// hit L2, 4 B written per triplet; a few hundred MB/s of HBM traffic at 1 G triplets/s.
#include "common.h"

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void philox_bpr_sample_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ sorted_items,
    const int32_t *__restrict__ row_user, int64_t n, uint32_t n_items, int shift, uint32_t seed_lo,
    uint32_t seed_hi, uint32_t epoch_lo, uint32_t epoch_hi, int32_t *__restrict__ j_out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t u = row_user[t];
        const int64_t b = indptr[u];
        const int32_t len = (int32_t)(indptr[u + 1] - b);
        const int32_t *row = sorted_items + b;
        uint32_t pick = 0;
        bool done = false;
        for (uint32_t block = 0; !done; block++) {
            uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), block, epoch_lo};
            philox4x32_10(c, seed_lo, seed_hi ^ epoch_hi);
#pragma unroll
            for (int w = 0; w < 4 && !done; w++) {
                const uint32_t r = c[w] >> shift;
                if (r >= n_items) continue;
                int32_t lo = 0, hi = len;  // lower_bound over the user's sorted positives
                while (lo < hi) {
                    const int32_t mid = (lo + hi) >> 1;
                    if ((uint32_t)row[mid] < r) lo = mid + 1; else hi = mid;
                }
                if (lo < len && (uint32_t)row[lo] == r) continue;
                pick = r; done = true;
            }
            if (block > 4096) { pick = 0xffffffffu; done = true; }  // every item positive
        }
        j_out[t] = (int32_t)pick;
    }
}

__global__ __launch_bounds__(256) void gather_pairs_kernel(const int32_t *__restrict__ perm, const int32_t *__restrict__ u,
                                                           const int32_t *__restrict__ i, int64_t n, int32_t *__restrict__ u_out,
                                                           int32_t *__restrict__ i_out) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t k = perm[t];
        u_out[t] = u[k];
        i_out[t] = i[k];
    }
}

}  // namespace

// rows of the training data in a device-drawn order: (u_out, i_out)[t] = (u, i)[perm[t]] -- the device twin of
// shuffle(self.data.trainingData) for the throughput mode of the pairwise models (base/deepRecommender.py:30)
extern "C" int qrec_gather_pairs(const int32_t *d_perm, const int32_t *d_u, const int32_t *d_i, int64_t n, int32_t *d_u_out,
                                 int32_t *d_i_out, void *stream) {
    QREC_REQUIRE(n >= 0, "qrec_gather_pairs: negative count");
    if (n == 0) return QREC_OK;
    QREC_REQUIRE(d_perm && d_u && d_i && d_u_out && d_i_out, "qrec_gather_pairs: null argument");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(gather_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, qrec::as_stream(stream), d_perm, d_u, d_i, n, d_u_out, d_i_out);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

extern "C" int qrec_philox_bpr_sample(const int64_t *d_indptr, const int32_t *d_sorted,
                                      const int32_t *d_row_user, int64_t n, int32_t n_items,
                                      uint64_t seed, uint64_t epoch, int32_t *d_j_out,
                                      void *stream) {
    QREC_REQUIRE(d_indptr && d_sorted && d_row_user && d_j_out && n >= 0 && n_items > 0,
                 "qrec_philox_bpr_sample: bad arguments");
    if (n == 0) return QREC_OK;
    const int shift = __builtin_clz((uint32_t)n_items);  // 32 - bit_length
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks/CU, grid-stride the rest
    hipLaunchKernelGGL(philox_bpr_sample_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       qrec::as_stream(stream), d_indptr, d_sorted, d_row_user, n,
                       (uint32_t)n_items, shift, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)epoch, (uint32_t)(epoch >> 32), d_j_out);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}
