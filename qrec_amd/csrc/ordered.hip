// Ordered scatter-add of rows: the deterministic counterpart of the float-atomic gradient scatters (parity mode).
//
//   out[key[s]] += ( sum over the slots s of class 0 with that key, in ascending s ) + ( ... class 1 ... ) + ...
//
// A "slot" is one looked-up row of a batch (tf.nn.embedding_lookup: LightGCN.py:22-24, BUIR.py:88-95, SEPT.py:239) and its
// contribution the gradient flowing back into that row; the gradient of a lookup is the sum of its slots' rows per table row,
// which TF (UnsortedSegmentSum on the CPU) and the fixtures' generator (index_put_ with accumulate) form sequentially in slot
// order, one lookup op (= class) at a time, and then add the ops' dense results.  Float atomics form the same sum in whatever
// order the memory system retires them: right to fp32 rounding, different from launch to launch.  Here:
//   1. the producing kernel writes slot s's contribution to contrib[s][ld] and its destination row to keys[s] (< 0: no row),
//   2. one stable radix sort of (key, slot) pairs (rocPRIM) groups the slots by row, ascending slot inside a row,
//   3. ordered_scatter_kernel: the group of lanes at the head of a run walks it, fp32 adds in that order, no contraction.
// Same inputs -> same bits, on every launch.  Cost at B = 2048, ld = 64: sort ~25 us + walk ~5 us: the parity mode's price,
// the throughput mode keeps the atomics.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

using namespace qrec;

namespace {

static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

int sort_temp_bytes(int64_t n, size_t *bytes) {
    static thread_local int64_t last_n = -1;
    static thread_local size_t last_bytes = 0;
    static thread_local int last_dev = -1;
    int dev = 0;
    QREC_HIP_CHECK(hipGetDevice(&dev));
    if (n == last_n && dev == last_dev) { *bytes = last_bytes; return QREC_OK; }
    size_t tb = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, (const int32_t *)nullptr, (int32_t *)nullptr,
                                                   rocprim::counting_iterator<int32_t>(0), (int32_t *)nullptr, (size_t)n, 0u, 32u, (hipStream_t)0);
    QREC_REQUIRE(e == hipSuccess, "ordered scatter: rocprim::radix_sort_pairs size query failed");
    *bytes = align256(tb ? tb : 256);
    last_n = n; last_bytes = *bytes; last_dev = dev;
    return QREC_OK;
}

template <int LPR>
__global__ __launch_bounds__(256) void ordered_scatter_kernel(const int32_t *__restrict__ keys, const int32_t *__restrict__ slots,
                                                              int64_t n, int64_t class_size, const float *__restrict__ contrib,
                                                              float *__restrict__ out) {
#pragma clang fp contract(off)
    constexpr int GPW = kWave / LPR, LD = 4 * LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, r = lane % LPR;
    for (int64_t p = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + g; p < n; p += (int64_t)gridDim.x * 4 * GPW) {
        const int32_t row = keys[p];
        if (row < 0 || (p > 0 && keys[p - 1] == row)) continue;          // no destination / not the head of its run
        f32x4 total = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
        int64_t cls = class_size ? slots[p] / class_size : 0;
        bool first = true;
        for (int64_t q = p; q < n && keys[q] == row; q++) {
            const int64_t s = slots[q];
            const int64_t c = class_size ? s / class_size : 0;
            if (c != cls) { total = first ? acc : total + acc; first = false; acc = f32x4{0.f, 0.f, 0.f, 0.f}; cls = c; }
            acc = acc + *reinterpret_cast<const f32x4 *>(contrib + s * LD + 4 * r);
        }
        total = first ? acc : total + acc;
        f32x4 *dst = reinterpret_cast<f32x4 *>(out + (int64_t)row * LD + 4 * r);
        *dst = *dst + total;
    }
}

}  // namespace

namespace qrec {

int ordered_ws_bytes(int64_t n_slots, int ld, int64_t *bytes) {
    size_t tb = 0;
    const int rc = sort_temp_bytes(n_slots > 0 ? n_slots : 1, &tb);
    if (rc != QREC_OK) return rc;
    *bytes = (int64_t)(align256((size_t)n_slots * ld * 4) + 3 * align256((size_t)n_slots * 4) + tb);
    return QREC_OK;
}

int ordered_ws_carve(void *ws, int64_t ws_bytes, int64_t n_slots, int ld, OrderedScatterWs *w) {
    int64_t need = 0;
    const int rc = ordered_ws_bytes(n_slots, ld, &need);
    if (rc != QREC_OK) return rc;
    QREC_REQUIRE(ws && ws_bytes >= need, "ordered scatter: workspace of %lld bytes, %lld needed (qrec_ordered_scatter_workspace_bytes)",
                 (long long)ws_bytes, (long long)need);
    QREC_REQUIRE(n_slots < ((int64_t)1 << 31), "ordered scatter: at most 2^31 - 1 slots");
    char *p = static_cast<char *>(ws);
    w->contrib = reinterpret_cast<float *>(p); p += align256((size_t)n_slots * ld * 4);
    w->keys = reinterpret_cast<int32_t *>(p); p += align256((size_t)n_slots * 4);
    w->keys_sorted = reinterpret_cast<int32_t *>(p); p += align256((size_t)n_slots * 4);
    w->slots_sorted = reinterpret_cast<int32_t *>(p); p += align256((size_t)n_slots * 4);
    w->temp = p;
    return sort_temp_bytes(n_slots > 0 ? n_slots : 1, &w->temp_bytes);
}

int ordered_scatter_run(const OrderedScatterWs &w, int64_t n_slots, int ld, int64_t class_size, float *out, hipStream_t st) {
    if (n_slots == 0) return QREC_OK;
    size_t tb = w.temp_bytes;
    // signed keys: rows < 0 (no destination) sort to the front and are skipped by the walk
    const hipError_t e = rocprim::radix_sort_pairs(w.temp, tb, (const int32_t *)w.keys, w.keys_sorted, rocprim::counting_iterator<int32_t>(0),
                                                   w.slots_sorted, (size_t)n_slots, 0u, 32u, st);
    QREC_REQUIRE(e == hipSuccess, "ordered scatter: rocprim::radix_sort_pairs failed");
    int64_t blocks;
#define QREC_OS(LPR)                                                                                                        \
    blocks = (n_slots + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)); if (blocks > 2048) blocks = 2048;                              \
    hipLaunchKernelGGL((ordered_scatter_kernel<LPR>), dim3((unsigned)blocks), dim3(256), 0, st, w.keys_sorted, w.slots_sorted, \
                       n_slots, class_size, w.contrib, out)
    switch (ld) {
        case 32: QREC_OS(8); break;
        case 64: QREC_OS(16); break;
        case 128: QREC_OS(32); break;
        case 256: QREC_OS(64); break;
        default: set_error("ordered scatter: row stride must be 32, 64, 128 or 256 floats (got %d)", ld); return QREC_ERR_INVALID;
    }
#undef QREC_OS
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}

}  // namespace qrec

extern "C" {

int qrec_ordered_scatter_workspace_bytes(int64_t n_slots, int32_t ld, int64_t *bytes) {
    QREC_REQUIRE(bytes && n_slots >= 0 && n_slots < ((int64_t)1 << 31) && ld > 0, "qrec_ordered_scatter_workspace_bytes: bad argument");
    return ordered_ws_bytes(n_slots, ld, bytes);
}

int qrec_scatter_add_rows_ordered(const float *d_src, const int32_t *d_dst_rows, int64_t n_slots, int32_t ld, int64_t class_size,
                                  float *d_out, void *d_workspace, int64_t workspace_bytes, void *stream) {
    QREC_REQUIRE(d_out && n_slots >= 0 && class_size >= 0, "qrec_scatter_add_rows_ordered: bad argument");
    QREC_REQUIRE(n_slots == 0 || (d_src && d_dst_rows), "qrec_scatter_add_rows_ordered: null source");
    QREC_REQUIRE(ld == 32 || ld == 64 || ld == 128 || ld == 256, "qrec_scatter_add_rows_ordered: row stride must be 32, 64, 128 or 256 floats (got %d)", ld);
    if (n_slots == 0) return QREC_OK;
    OrderedScatterWs w;
    const int rc = ordered_ws_carve(d_workspace, workspace_bytes, n_slots, ld, &w);
    if (rc != QREC_OK) return rc;
    hipStream_t st = as_stream(stream);
    // the caller's rows are used where they lie: only the keys are copied next to the sort's buffers
    QREC_HIP_CHECK(hipMemcpyAsync(w.keys, d_dst_rows, sizeof(int32_t) * (size_t)n_slots, hipMemcpyDeviceToDevice, st));
    OrderedScatterWs v = w;
    v.contrib = const_cast<float *>(d_src);
    return ordered_scatter_run(v, n_slots, ld, class_size, d_out, st);
}

}  // extern "C"
