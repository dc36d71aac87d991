// Host side of the order-exact BPR mode beyond one wavefront (model/ranking/BPR.py:28-53).
//
// The reference applies the epoch's triplets one after another; triplet t reads and rewrites the rows P[u_t], Q[i_t],
// Q[j_t], so it depends on the previous toucher of each of the three -- nothing else.  Sampling does not look at the
// embeddings, so once the negatives are drawn the whole dependence DAG is known before the first update.  This file
// turns it into a STATIC schedule: device time steps of at most `width` mutually independent triplets, every triplet
// no earlier than one step after the last toucher of each of its rows (list scheduling in the reference's order, so a
// triplet whose rows are free overtakes its stalled predecessors -- which changes no value: all conflicts are ordered).
// For each row of each scheduled triplet the schedule also says where the consumer finds the current value: in the
// on-chip forwarding buffer of one of the last two steps (which step, which slot, which of its rows), or in the
// table in memory (last toucher at least three steps back: its store is visible to a load issued two steps ahead of the use; see bpr_exact.hip).
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

using namespace qrec;

namespace {
struct NextFree {   // steps that still have a free slot: union-find "next step >= s with room"
    std::vector<int32_t> parent;
    int32_t find(int32_t s) {
        if ((size_t)s >= parent.size()) {
            size_t old = parent.size();
            parent.resize(std::max<size_t>(s + 1, old * 2 + 16));
            for (size_t k = old; k < parent.size(); ++k) parent[k] = (int32_t)k;
        }
        int32_t r = s;
        while (parent[r] != r) r = parent[r];
        while (parent[s] != r) { int32_t nx = parent[s]; parent[s] = r; s = nx; }
        return r;
    }
    void fill(int32_t s) { find(s + 1); parent[s] = s + 1; }
};
}  // namespace

extern "C" int qrec_bpr_exact_schedule(const int32_t *h_u, const int32_t *h_i, const int32_t *h_j, int64_t n,
                                       int32_t n_users, int32_t n_items, int32_t width, int32_t *h_entries,
                                       int32_t *h_step_off, int64_t *n_steps_out) {
    QREC_REQUIRE(n >= 0 && n < (1ll << 31) && n_users >= 0 && n_items >= 0, "qrec_bpr_exact_schedule: bad sizes");
    QREC_REQUIRE(width >= 1 && width <= QREC_EXACT_MAX_WIDTH, "qrec_bpr_exact_schedule: width must be in 1..%d", QREC_EXACT_MAX_WIDTH);
    QREC_REQUIRE(n_steps_out && h_step_off && (n == 0 || (h_u && h_i && h_j && h_entries)), "qrec_bpr_exact_schedule: null argument");
    struct Last { int32_t step = -1000, slot = 0, which = 0; };
    std::vector<Last> last((size_t)n_users + (size_t)n_items);
    std::vector<int32_t> step_of(n), slot_of(n), count;
    std::vector<int32_t> src(3 * (size_t)n);
    NextFree nf;
    int32_t n_steps = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t u = h_u[t], i = h_i[t], j = h_j[t];
        QREC_REQUIRE(u >= 0 && u < n_users && i >= 0 && i < n_items && j >= 0 && j < n_items && i != j,
                     "qrec_bpr_exact_schedule: triplet %lld out of range (u=%d i=%d j=%d)", (long long)t, u, i, j);
        const size_t rows[3] = {(size_t)u, (size_t)n_users + i, (size_t)n_users + j};
        int32_t earliest = 0;
        for (size_t r : rows) earliest = std::max(earliest, last[r].step + 1);
        const int32_t s = nf.find(earliest);
        if ((size_t)s >= count.size()) count.resize(std::max<size_t>(s + 1, count.size() * 2 + 16), 0);
        const int32_t slot = count[s]++;
        if (count[s] == width) nf.fill(s);
        step_of[t] = s; slot_of[t] = slot;
        n_steps = std::max(n_steps, s + 1);
        for (int k = 0; k < 3; ++k) {
            Last &l = last[rows[k]];
            const int32_t dist = s - l.step;                      // >= 1
            src[3 * t + k] = dist <= 2 ? ((dist - 1) * QREC_EXACT_MAX_WIDTH + l.slot) * 3 + l.which : -1;
            l.step = s; l.slot = slot; l.which = k;
        }
    }
    // step-major layout: entries of step s are h_entries[8*h_step_off[s] ...), slot order
    std::fill(h_step_off, h_step_off + n_steps + 1, 0);
    for (int64_t t = 0; t < n; ++t) h_step_off[step_of[t] + 1]++;
    for (int32_t s = 0; s < n_steps; ++s) h_step_off[s + 1] += h_step_off[s];
    for (int64_t t = 0; t < n; ++t) {
        int32_t *e = h_entries + 8 * ((int64_t)h_step_off[step_of[t]] + slot_of[t]);
        e[0] = h_u[t]; e[1] = h_i[t]; e[2] = h_j[t]; e[3] = (int32_t)t;
        e[4] = src[3 * t]; e[5] = src[3 * t + 1]; e[6] = src[3 * t + 2]; e[7] = 0;
    }
    *n_steps_out = n_steps;
    return QREC_OK;
}
