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
            src[3 * t + k] = dist <= 2 ? ((dist - 1) * QREC_EXACT_MAX_WIDTH + l.slot) * 4 + l.which : -1;   // fields on bit boundaries: the kernels decode with shifts
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

// Round 3, the schedule of the four-triplets-per-wavefront kernel (bpr_levels_reg_kernel): a row reaches its next toucher either
// through the REGISTERS of the slot that wrote it -- P[u] along a user's run: the run stays on one slot, consecutive steps --
// or through the table, whose copy is current for loads issued two steps ahead once the last toucher is >= 3 steps back.
// Nothing goes through LDS, so a triplet whose rows were touched one or two steps ago by ANOTHER slot waits until they are
// three steps old: 228.6 k steps instead of 214.4 k at the Yelp2018 shape (width 8; 85 % of the P rows ride in registers), each
// step without the LDS write -> barrier -> LDS read turn-around on its dependent chain.  Same order, same values.
// Slots of a step are no longer dense (a run keeps ITS slot): entry word 7 = handed-on bit 0 (P goes on in registers: no table
// store) | slot << 8 | 0x1000 (marks this format for qrec_bpr_exact_expand).  src_P = -2: registers; every other source -1.
extern "C" int qrec_bpr_exact_schedule_reg(const int32_t *h_u, const int32_t *h_i, const int32_t *h_j, int64_t n, int32_t n_users,
                                           int32_t n_items, int32_t width, int32_t *h_entries, int32_t *h_step_off,
                                           int64_t *n_steps_out) {
    QREC_REQUIRE(n >= 0 && n < (1ll << 31) && n_users >= 0 && n_items >= 0, "qrec_bpr_exact_schedule_reg: bad sizes");
    QREC_REQUIRE(width >= 1 && width <= QREC_EXACT_MAX_WIDTH, "qrec_bpr_exact_schedule_reg: width must be in 1..%d", QREC_EXACT_MAX_WIDTH);
    QREC_REQUIRE(n_steps_out && h_step_off && (n == 0 || (h_u && h_i && h_j && h_entries)), "qrec_bpr_exact_schedule_reg: null argument");
    struct Last { int32_t step = -1000, slot = 0; int64_t t = -1; };
    std::vector<Last> last((size_t)n_users + (size_t)n_items);
    std::vector<int32_t> step_of(n), slot_of(n);
    std::vector<uint8_t> from_reg(n, 0), handed_on(n, 0);
    std::vector<uint16_t> used;                         // per step: bit k = slot k taken
    auto ensure = [&](size_t s) { if (s >= used.size()) used.resize(std::max(s + 1, used.size() * 2 + 16), 0); };
    const uint16_t full = (uint16_t)((1u << width) - 1u);
    NextFree nf;                                        // next step >= s that still has a free slot
    int32_t n_steps = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int32_t u = h_u[t], i = h_i[t], j = h_j[t];
        QREC_REQUIRE(u >= 0 && u < n_users && i >= 0 && i < n_items && j >= 0 && j < n_items && i != j,
                     "qrec_bpr_exact_schedule_reg: triplet %lld out of range (u=%d i=%d j=%d)", (long long)t, u, i, j);
        Last &lp = last[(size_t)u], &li = last[(size_t)n_users + i], &lj = last[(size_t)n_users + j];
        const int32_t eq = std::max(0, std::max(li.step, lj.step) + 3);
        int32_t s = std::max(eq, lp.step + 1), slot = -1;
        for (;;) {
            s = nf.find(s);
            ensure((size_t)s);
            const int32_t dp = s - lp.step;
            if (dp == 1) {
                if (!((used[s] >> lp.slot) & 1)) { slot = lp.slot; break; }      // the run goes on in its slot's registers
                s += 2; continue;                                               // its slot is taken: wait for the table copy
            }
            if (dp == 2) { s += 1; continue; }
            break;
        }
        const bool reg = slot >= 0;
        if (!reg) { slot = 0; while ((used[s] >> slot) & 1) ++slot; }
        used[s] |= (uint16_t)(1u << slot);
        if (used[s] == full) nf.fill(s);
        step_of[t] = s; slot_of[t] = slot;
        n_steps = std::max(n_steps, s + 1);
        if (reg) { from_reg[t] = 1; handed_on[lp.t] = 1; }
        lp.step = li.step = lj.step = s; lp.slot = li.slot = lj.slot = slot; lp.t = li.t = lj.t = t;
    }
    std::fill(h_step_off, h_step_off + n_steps + 1, 0);
    for (int64_t t = 0; t < n; ++t) h_step_off[step_of[t] + 1]++;
    for (int32_t s = 0; s < n_steps; ++s) h_step_off[s + 1] += h_step_off[s];
    std::vector<int32_t> fill(h_step_off, h_step_off + n_steps);
    for (int64_t t = 0; t < n; ++t) {       // triplets of a step in program order (slots are carried in word 7)
        int32_t *e = h_entries + 8 * (int64_t)fill[step_of[t]]++;
        e[0] = h_u[t]; e[1] = h_i[t]; e[2] = h_j[t]; e[3] = (int32_t)t;
        e[4] = from_reg[t] ? -2 : -1; e[5] = -1; e[6] = -1; e[7] = (int32_t)handed_on[t] | (slot_of[t] << 8) | 0x1000;
    }
    *n_steps_out = n_steps;
    return QREC_OK;
}
