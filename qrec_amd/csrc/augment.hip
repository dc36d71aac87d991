// Per-epoch graph augmentation on the device (throughput mode of SGL and BUIR).
//
// Reference: SGL._create_adj_mat (model/ranking/SGL.py:113-155), BUIR.get_adj_mat (model/ranking/BUIR.py:41-65): every epoch,
// per view (and per layer for the random-walk variant), a sub-graph of the training graph is drawn --
//     node dropout (aug 0):        random.sample of int(U rate) users and of int(I rate) items; an edge survives iff both ends do
//     edge dropout / random walk:  random.sample of int(E (1 - rate)) training rows
// -- and re-normalised: A' = R' + R'^T, d' = rowsum(A')^-1/2 (inf -> 0), value = fl32(fl32(d'_r a') d'_c), as scipy CSR.
// The exact mode replays CPython's random.sample on the host and rebuilds CSR + launch plan there (12 % of an SGL epoch at the
// Yelp2018 shape, round 5).  Here nothing leaves the device and nothing is rebuilt:
//
//   * a sub-graph's non-zeros are a SUBSET of the full graph's, so its matrix is the full graph's CSR structure -- the one the
//     SpMM plan (segments, XCD dealing) was built for, once -- with another VALUE array: dropped entries are 0, kept entries
//     carry the sub-graph's own normalisation.  Adding a +0 term changes no sum, so a product over this array equals the product
//     over the compacted sub-graph up to the association of long rows' partial sums.
//   * the draw: a uniformly random subset of exact size K = the first K entries of a uniformly random permutation
//     (qrec_random_permutations: Philox4x32-10 keys, one stable radix sort) -- the distribution of random.sample's subset, from the
//     counter-based stream of the throughput mode (restated on the CPU by the oracle: tests hold the device stream to it bit for bit).
//   * subgraph_count_kernel   kept training row t -> cnt[pos_ui[t]] += 1, cnt[pos_iu[t]] += 1 (its two CSR entries; duplicated
//                             training rows add up, as in the reference's csr_matrix), deg[u] += 1, deg[U + i] += 1   (int atomics: exact)
//   * subgraph_values_kernel  value[e] = cnt[e] ? fl32(fl32(dinv[deg[row e]] * cnt[e]) * dinv[deg[col e]]) : 0, dinv[] a host-made
//                             table of numpy's own float32 power(k, -0.5) for k = 0 .. max degree: bit-identical to the reference's values.
// Bytes per draw at the Yelp2018 shape: sort of 1.24 M keys + 2 x 1.24 M int atomics + 2.47 M values written: ~0.15 ms, against
// 14 ms of host work per sub-graph in round 5.
#include "common.h"

using namespace qrec;

namespace {

__global__ void mark_ids_kernel(const int32_t *__restrict__ ids, int64_t n, uint8_t *__restrict__ flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) flags[ids[k]] = 1;
}

// keep_rows != nullptr: the kept training rows (edge dropout / random walk); else every row whose end points are not flagged
__global__ void subgraph_count_kernel(const int32_t *__restrict__ u, const int32_t *__restrict__ i, const int32_t *__restrict__ pos_ui,
                                      const int32_t *__restrict__ pos_iu, int64_t n_edges, int n_users,
                                      const int32_t *__restrict__ keep_rows, int64_t n_keep, const uint8_t *__restrict__ drop_user,
                                      const uint8_t *__restrict__ drop_item, int32_t *__restrict__ cnt, int32_t *__restrict__ deg) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t t;
    if (keep_rows) {
        if (k >= n_keep) return;
        t = keep_rows[k];
    } else {
        if (k >= n_edges) return;
        t = k;
    }
    const int uu = u[t], ii = i[t];
    if (drop_user && (drop_user[uu] || drop_item[ii])) return;
    atomicAdd(cnt + pos_ui[t], 1); atomicAdd(cnt + pos_iu[t], 1);
    atomicAdd(deg + uu, 1); atomicAdd(deg + n_users + ii, 1);
}

__global__ void subgraph_values_kernel(const int32_t *__restrict__ cnt, const int32_t *__restrict__ deg, const int32_t *__restrict__ row_of,
                                       const int32_t *__restrict__ col_of, int64_t nnz, const float *__restrict__ dinv, int max_deg,
                                       float *__restrict__ values) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int c = cnt[e];
    float v = 0.f;
    if (c) {
        const int dr = deg[row_of[e]], dc = deg[col_of[e]];
        // degrees beyond the table cannot occur (a sub-graph's degrees are bounded by the full graph's); clamp instead of reading past it
        const float a = dinv[dr < max_deg ? dr : max_deg], b = dinv[dc < max_deg ? dc : max_deg];
        v = (a * (float)c) * b;
    }
    values[e] = v;
}

}  // namespace

extern "C" {

int qrec_subgraph_values(const int32_t *d_u, const int32_t *d_i, const int32_t *d_pos_ui, const int32_t *d_pos_iu, int64_t n_edges,
                         int32_t n_users, int32_t n_items, const int32_t *d_keep_rows, int64_t n_keep, const int32_t *d_drop_users,
                         int64_t n_drop_users, const int32_t *d_drop_items, int64_t n_drop_items, const int32_t *d_row_of_nnz,
                         const int32_t *d_indices, int64_t nnz, const float *d_dinv_table, int32_t max_deg, int32_t *d_cnt,
                         int32_t *d_deg, uint8_t *d_flags, float *d_values, void *stream) {
    QREC_REQUIRE(d_u && d_i && d_pos_ui && d_pos_iu && d_row_of_nnz && d_indices && d_dinv_table && d_cnt && d_deg && d_values,
                 "qrec_subgraph_values: null argument");
    QREC_REQUIRE(n_edges >= 0 && n_edges < ((int64_t)1 << 31) && nnz >= 0 && nnz < ((int64_t)1 << 31) && n_users >= 0 && n_items >= 0 && max_deg >= 0,
                 "qrec_subgraph_values: bad sizes");
    QREC_REQUIRE(n_keep >= 0 && n_keep <= n_edges, "qrec_subgraph_values: bad keep list");
    const bool nodes = d_drop_users || d_drop_items;
    QREC_REQUIRE(!(nodes && d_keep_rows), "qrec_subgraph_values: either a list of kept rows (edge dropout) or dropped nodes (node dropout), not both");
    QREC_REQUIRE(!nodes || (d_flags && n_drop_users >= 0 && n_drop_users <= n_users && n_drop_items >= 0 && n_drop_items <= n_items &&
                            (n_drop_users == 0 || d_drop_users) && (n_drop_items == 0 || d_drop_items)),
                 "qrec_subgraph_values: node dropout needs the flag scratch (n_users + n_items bytes) and both id lists");
    hipStream_t st = as_stream(stream);
    const int64_t n_nodes = (int64_t)n_users + n_items;
    QREC_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t)nnz, st));
    QREC_HIP_CHECK(hipMemsetAsync(d_deg, 0, sizeof(int32_t) * (size_t)n_nodes, st));
    const uint8_t *drop_u = nullptr, *drop_i = nullptr;
    if (nodes) {
        QREC_HIP_CHECK(hipMemsetAsync(d_flags, 0, (size_t)n_nodes, st));
        if (n_drop_users) hipLaunchKernelGGL(mark_ids_kernel, dim3((unsigned)((n_drop_users + 255) / 256)), dim3(256), 0, st, d_drop_users, n_drop_users, d_flags);
        if (n_drop_items) hipLaunchKernelGGL(mark_ids_kernel, dim3((unsigned)((n_drop_items + 255) / 256)), dim3(256), 0, st, d_drop_items, n_drop_items, d_flags + n_users);
        QREC_LAUNCH_CHECK();
        drop_u = d_flags; drop_i = d_flags + n_users;
    }
    const int64_t work = d_keep_rows ? n_keep : n_edges;
    if (work > 0) {
        hipLaunchKernelGGL(subgraph_count_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, d_u, d_i, d_pos_ui, d_pos_iu, n_edges,
                           n_users, d_keep_rows, n_keep, drop_u, drop_i, d_cnt, d_deg);
        QREC_LAUNCH_CHECK();
    }
    if (nnz > 0) {
        hipLaunchKernelGGL(subgraph_values_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, d_cnt, d_deg, d_row_of_nnz, d_indices,
                           nnz, d_dinv_table, max_deg, d_values);
        QREC_LAUNCH_CHECK();
    }
    return QREC_OK;
}

}  // extern "C"
