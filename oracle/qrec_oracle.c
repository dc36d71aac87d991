/*
 * qrec_oracle.c -- CPU restatement of QRec's embedding-training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker*: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * path (qrec_amd/, libqrec_hip.so) never links, imports or calls anything in oracle/.
 *
 * Parity pinning: the reference (Coder-Yu/QRec) has no tests or golden vectors of its
 * own, so this restatement is pinned against outputs of the reference's
 * own numpy path, run in-process from /root/reference by tests/golden/gen_golden.py and
 * committed under tests/golden/ (index streams bit-exact, fp64 state to ~1e-12).
 *
 * Every function cites the reference file:line it restates.  Third-party algorithms
 * that the reference relies on but does not vendor:
 *   - CPython 3.10 `random` (Lib/random.py, Modules/_randommodule.c): MT19937,
 *     init_by_array seeding, getrandbits, _randbelow_with_getrandbits, choice, shuffle,
 *     random().
 *   - numpy legacy RandomState (numpy/random/_mt19937.pyx, mtrand.pyx): init_genrand
 *     seeding, random_sample.
 * Both are restated from their published algorithms and verified against the live
 * interpreters in tests/test_oracle_rng.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t mt[MT_N];
    int32_t pos; /* 0..624 ; 624 = regenerate on next draw (CPython "index") */
} orc_mt;

/* ---- MT19937 core (Matsumoto & Nishimura 2002; CPython _randommodule.c genrand_uint32) */
static void mt_regen(orc_mt *s) {
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    uint32_t *mt = s->mt, y;
    int kk;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
    }
    for (; kk < MT_N - 1; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
    s->pos = 0;
}

static inline uint32_t mt_u32(orc_mt *s) {
    uint32_t y;
    if (s->pos >= MT_N) mt_regen(s);
    y = s->mt[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static void mt_init_genrand(orc_mt *s, uint32_t seed) {
    int i;
    s->mt[0] = seed;
    for (i = 1; i < MT_N; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = MT_N;
}

static void mt_init_by_array(orc_mt *s, const uint32_t *key, int len) {
    int i = 1, j = 0, k;
    uint32_t *mt = s->mt;
    mt_init_genrand(s, 19650218u);
    k = (MT_N > len ? MT_N : len);
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
    s->pos = MT_N;
}

/* CPython random.seed(int a): key = little-endian 32-bit words of abs(a), >= 1 word. */
void orc_seed_cpython(orc_mt *s, uint64_t a) {
    uint32_t key[2];
    int len = 1;
    key[0] = (uint32_t)(a & 0xffffffffu);
    key[1] = (uint32_t)(a >> 32);
    if (key[1]) len = 2;
    mt_init_by_array(s, key, len);
}

/* numpy np.random.seed(int s) (legacy seeding): init_genrand(s). */
void orc_seed_numpy(orc_mt *s, uint32_t seed) { mt_init_genrand(s, seed); }

/* random.getstate()[1] / setstate interop: 624 words + index. */
void orc_set_state(orc_mt *s, const uint32_t *words625) {
    memcpy(s->mt, words625, MT_N * sizeof(uint32_t));
    s->pos = (int32_t)words625[MT_N];
}
void orc_get_state(const orc_mt *s, uint32_t *words625) {
    memcpy(words625, s->mt, MT_N * sizeof(uint32_t));
    words625[MT_N] = (uint32_t)s->pos;
}
int orc_state_size(void) { return (int)sizeof(orc_mt); }

uint32_t orc_u32(orc_mt *s) { return mt_u32(s); }

/* CPython random.random() / numpy random_sample(): 53-bit double from two words. */
static inline double mt_double(orc_mt *s) {
    uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}
double orc_random(orc_mt *s) { return mt_double(s); }

/* numpy np.random.rand(n) -> n doubles (used as rand(U,d)/3 by
 * base/iterativeRecommender.py:37-38). */
void orc_numpy_rand(orc_mt *s, double *out, int64_t n) {
    int64_t k;
    for (k = 0; k < n; k++) out[k] = mt_double(s);
}

static inline int bit_length_u32(uint32_t n) {
    int k = 0;
    while (n) { k++; n >>= 1; }
    return k;
}

/* CPython Random._randbelow_with_getrandbits(n), n < 2^32:
 * k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
 * getrandbits(k<=32) = genrand_uint32() >> (32-k). */
static inline uint32_t mt_randbelow(orc_mt *s, uint32_t n, int k) {
    uint32_t r = mt_u32(s) >> (32 - k);
    while (r >= n) r = mt_u32(s) >> (32 - k);
    return r;
}
uint32_t orc_randbelow(orc_mt *s, uint32_t n) {
    if (!n) return 0;
    return mt_randbelow(s, n, bit_length_u32(n));
}

/* random.shuffle(x) (Lib/random.py): for i in reversed(range(1,len(x))):
 * j = randbelow(i+1); x[i],x[j] = x[j],x[i].   Called once per epoch by
 * base/iterativeRecommender.py:101 (isConverged) and base/deepRecommender.py:30.
 * perm may be NULL: then only the generator is advanced. */
void orc_shuffle(orc_mt *s, int64_t *perm, int64_t n) {
    int64_t i;
    for (i = n - 1; i >= 1; i--) {
        uint32_t j = mt_randbelow(s, (uint32_t)(i + 1), bit_length_u32((uint32_t)(i + 1)));
        if (perm) { int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    }
}

/* random.sample(range(n), k) (Lib/random.py, CPython 3.10): used by model/ranking/SGL.py:118-135 to
 * pick the kept edges / dropped nodes of an augmented sub-graph.
 *   setsize = 21; if k > 5: setsize += 4 ** ceil(log(k*3, 4))
 *   if n <= setsize:  pool = list(range(n)); for i in range(k): j = randbelow(n-i); out[i] = pool[j];
 *                     pool[j] = pool[n-i-1]
 *   else:             selected = set(); for i in range(k): j = randbelow(n); while j in selected: redraw;
 *                     add j; out[i] = j                                                               */
static int64_t sample_setsize(int64_t k) {
    int64_t setsize = 21;
    if (k > 5) {
        /* 4 ** ceil(log(k*3, 4)) with Python's float log: smallest power of 4 >= 3k, computed the same way */
        double e = ceil(log((double)k * 3.0) / log(4.0));
        setsize += (int64_t)pow(4.0, e);
    }
    return setsize;
}
void orc_sample_range(orc_mt *s, int64_t n, int64_t k, int64_t *out) {
    int64_t i;
    if (n <= sample_setsize(k)) {
        int64_t *pool = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        for (i = 0; i < n; i++) pool[i] = i;
        for (i = 0; i < k; i++) {
            uint32_t m = (uint32_t)(n - i);
            uint32_t j = mt_randbelow(s, m, bit_length_u32(m));
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
        free(pool);
    } else {
        uint8_t *sel = (uint8_t *)calloc((size_t)n, 1);
        int kb = bit_length_u32((uint32_t)n);
        for (i = 0; i < k; i++) {
            uint32_t j = mt_randbelow(s, (uint32_t)n, kb);
            while (sel[j]) j = mt_randbelow(s, (uint32_t)n, kb);
            sel[j] = 1; out[i] = j;
        }
        free(sel);
    }
}

/* util/dataSplit.py:9-26 DataSplit.dataSplit: one random() per row; row goes to the
 * test set iff random() < test_ratio.  (binarized rows have rating 1 -> kept.) */
void orc_data_split(orc_mt *s, int64_t n, double test_ratio, uint8_t *is_test) {
    int64_t k;
    if (test_ratio >= 1 || test_ratio <= 0) test_ratio = 0.3;
    for (k = 0; k < n; k++) is_test[k] = mt_double(s) < test_ratio;
}

/* ------------------------------------------------------------------------------------
 * a-1  Triplet sampler, numpy BPR path: model/ranking/BPR.py:28-38.
 *   for user in PositiveSet (id order): for item in PositiveSet[user] (row order):
 *       item_j = choice(itemList); while item_j in PositiveSet[user]: redraw
 *   itemList = list(data.item.keys()) -> index r is item id r.
 * pos_indptr/pos_indices: CSR of PositiveSet (users in id order, items in dict order).
 * Writes one j per CSR entry.  Returns the number of MT words consumed (diagnostic).
 */
int64_t orc_bpr_sample_epoch(orc_mt *s, const int64_t *pos_indptr, const int32_t *pos_indices,
                             int32_t n_users, int32_t n_items, int32_t *j_out) {
    int32_t *stamp = (int32_t *)calloc((size_t)n_items, sizeof(int32_t));
    int k = bit_length_u32((uint32_t)n_items);
    int64_t words = 0;
    int32_t u;
    for (u = 0; u < n_users; u++) {
        int64_t e, b = pos_indptr[u], en = pos_indptr[u + 1];
        for (e = b; e < en; e++) stamp[pos_indices[e]] = u + 1;
        for (e = b; e < en; e++) {
            uint32_t r;
            for (;;) {
                r = mt_u32(s) >> (32 - k); words++;
                if (r >= (uint32_t)n_items) continue; /* _randbelow redraw */
                if (stamp[r] == u + 1) continue;      /* BPR.py:36-37 redraw */
                break;
            }
            j_out[e] = (int32_t)r;
        }
    }
    free(stamp);
    return words;
}

/* a-1, throughput mode: the counter-based negative sampler of include/qrec_hip.h (qrec_philox_bpr_sample).  NOT the
 * reference's stream (that is orc_bpr_sample_epoch above) -- the same DISTRIBUTION as BPR.py:35-37 (uniform over the items
 * that are not positives of the user, by rejection) drawn so that a triplet's negative depends on (seed, epoch, t) alone.
 * Third-party algorithm, restated from its publication: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
 * numbers: as easy as 1, 2, 3", SC'11; Random123 1.x philox.h) -- multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key bumps
 * 0x9E3779B9 / 0xBB67AE85, ten rounds; pinned to Random123's known-answer vectors in tests/test_oracle_rng.py.
 * Contract of the sampler (the device kernel is held to this function bit for bit):
 *   counter = {t_lo, t_hi, block, epoch_lo}, key = {seed_lo, seed_hi ^ epoch_hi}, block = 0, 1, ...
 *   the block's four words in order: candidate = word >> (32 - bit_length(n_items)); skipped if >= n_items or if it is a
 *   positive of row_user[t] (sorted CSR row); the first survivor is j[t].  None in blocks 0 .. 4096: -1 (every item positive). */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    int r;
    for (r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
void orc_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof c);
}
static int row_contains(const int32_t *a, int64_t n, int32_t x);
void orc_philox_bpr_sample(const int64_t *indptr, const int32_t *sorted_items, const int32_t *row_user, int64_t n,
                           int32_t n_items, uint64_t seed, uint64_t epoch, int32_t *j_out) {
    const int shift = 32 - bit_length_u32((uint32_t)n_items);
    int64_t t;
    for (t = 0; t < n; t++) {
        const int32_t u = row_user[t];
        const int32_t *row = sorted_items + indptr[u];
        const int64_t len = indptr[u + 1] - indptr[u];
        int32_t pick = -1;
        uint32_t block;
        for (block = 0; pick < 0 && block <= 4096; block++) {
            uint32_t c[4] = {(uint32_t)t, (uint32_t)((uint64_t)t >> 32), block, (uint32_t)epoch};
            int w;
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(epoch >> 32));
            for (w = 0; w < 4 && pick < 0; w++) {
                const uint32_t r = shift ? (c[w] >> shift) : c[w];
                if (r >= (uint32_t)n_items || row_contains(row, len, (int32_t)r)) continue;
                pick = (int32_t)r;
            }
        }
        j_out[t] = pick;
    }
}

/* f-2, throughput mode: the sort keys of the device's uniform permutations (include/qrec_hip.h, qrec_random_permutations with
 * count = 1; csrc/mhcn.hip perm_keys_kernel): key[i] = the top 40 bits of the first two Philox words of
 * counter {i_lo, i_hi, stream_lo, stream_hi}, key {seed_lo, seed_hi}.  The permutation is the STABLE argsort of the keys (equal
 * keys -- 2^-40 per pair -- keep index order); a sub-graph of SGL / BUIR keeps / drops the first K entries of it
 * (SGL.py:118-130's random.sample subsets, drawn from this stream instead of CPython's). */
void orc_philox_perm_keys(int64_t n, uint64_t seed, uint64_t stream_id, uint64_t *keys_out) {
    int64_t i;
    for (i = 0; i < n; i++) {
        uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        keys_out[i] = (((uint64_t)c[0] << 32) | c[1]) >> 24;
    }
}

/* a-2  Triplet sampler, TF path: base/deepRecommender.py:29-52 (next_batch_pairwise).
 * The caller applies shuffle(trainingData) first (orc_shuffle on a row permutation);
 * this function then walks the rows in the shuffled order and draws one negative per
 * row: neg = choice(item_list); while neg in trainSet_u[user]: redraw.  The batch
 * boundaries do not touch the generator, so one call covers the whole epoch.
 * rated_indptr/rated_indices: CSR of trainSet_u (ALL train items of the user),
 * must be sorted within each row (membership by binary search). */
static int row_contains(const int32_t *a, int64_t n, int32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == x;
}
void orc_pairwise_sample_epoch(orc_mt *s, const int32_t *row_user, int64_t n_rows,
                               const int64_t *rated_indptr, const int32_t *rated_sorted,
                               int32_t n_items, int32_t *neg_out) {
    int k = bit_length_u32((uint32_t)n_items);
    int64_t t;
    for (t = 0; t < n_rows; t++) {
        int32_t u = row_user[t];
        const int32_t *row = rated_sorted + rated_indptr[u];
        int64_t len = rated_indptr[u + 1] - rated_indptr[u];
        uint32_t r;
        for (;;) {
            r = mt_u32(s) >> (32 - k);
            if (r >= (uint32_t)n_items) continue;
            if (row_contains(row, len, (int32_t)r)) continue;
            break;
        }
        neg_out[t] = (int32_t)r;
    }
}

/* ------------------------------------------------------------------------------------
 * a-5  BPR SGD step, numpy path: model/ranking/BPR.py:45-53 (optimization) and
 * util/qmath.py:127-128 (sigmoid = 1/(1+exp(-x))).  Strictly sequential, in place:
 *   s = sigmoid(P[u].Q[i] - P[u].Q[j])                       (:46)
 *   P[u] += lr*(1-s)*(Q[i]-Q[j])                             (:47)
 *   Q[i] += lr*(1-s)*P[u]      (uses the UPDATED P[u])       (:48)
 *   Q[j] -= lr*(1-s)*P[u]                                    (:49)
 *   P[u] -= lr*regU*P[u]; Q[i] -= lr*regI*Q[i]; Q[j] -= lr*regI*Q[j]   (:50-52)
 *   loss += -log(s)                                          (:53)
 * Returns sum of -log(s) over the n triplets (the epoch-end reg term is orc_sumsq). */
double orc_bpr_sgd_f64(double *P, double *Q, int32_t d, const int32_t *u_idx,
                       const int32_t *i_idx, const int32_t *j_idx, int64_t n,
                       double lr, double regU, double regI) {
    double loss = 0.0;
    int64_t t;
    int c;
    for (t = 0; t < n; t++) {
        double *pu = P + (int64_t)u_idx[t] * d;
        double *qi = Q + (int64_t)i_idx[t] * d;
        double *qj = Q + (int64_t)j_idx[t] * d;
        double xi = 0.0, xj = 0.0, s, g;
        for (c = 0; c < d; c++) { xi += pu[c] * qi[c]; xj += pu[c] * qj[c]; }
        s = 1.0 / (1.0 + exp(-(xi - xj)));
        g = lr * (1.0 - s);
        for (c = 0; c < d; c++) pu[c] += g * (qi[c] - qj[c]);
        for (c = 0; c < d; c++) qi[c] += g * pu[c];
        for (c = 0; c < d; c++) qj[c] -= g * pu[c];
        for (c = 0; c < d; c++) pu[c] -= (lr * regU) * pu[c];
        for (c = 0; c < d; c++) qi[c] -= (lr * regI) * qi[c];
        for (c = 0; c < d; c++) qj[c] -= (lr * regI) * qj[c];
        loss += -log(s);
    }
    return loss;
}

/* ------------------------------------------------------------------------------------
 * TBPR (model/ranking/TBPR.py:111-166): per user (PositiveSet order) and positive item the chain
 * [i, choice(jointItems)?, choice(weakItems)?, choice(strongItems)?, k] with k = choice(itemList) redrawn while positive
 * (:137-155); consecutive members form the (u, a, b) updates of TBPR.optimization (:157-158).  Lists are CSR over users.
 * Returns the number of triplets written (<= 4 per positive).
 */
int64_t orc_tbpr_sample_epoch(orc_mt *s, const int64_t *pos_indptr, const int32_t *pos_items, int32_t n_users, int32_t n_items,
                              const int64_t *j_ptr, const int32_t *j_items, const int64_t *w_ptr, const int32_t *w_items,
                              const int64_t *s_ptr, const int32_t *s_items, int32_t *u_out, int32_t *a_out, int32_t *b_out) {
    int32_t *stamp = (int32_t *)calloc((size_t)n_items, sizeof(int32_t));
    const int kbits = bit_length_u32((uint32_t)n_items);
    int64_t n = 0, e;
    int32_t u;
    const int64_t *ptrs[3];
    const int32_t *lists[3];
    ptrs[0] = j_ptr; ptrs[1] = w_ptr; ptrs[2] = s_ptr; lists[0] = j_items; lists[1] = w_items; lists[2] = s_items;
    for (u = 0; u < n_users; u++) {
        for (e = pos_indptr[u]; e < pos_indptr[u + 1]; e++) stamp[pos_items[e]] = u + 1;
        for (e = pos_indptr[u]; e < pos_indptr[u + 1]; e++) {
            int32_t chain[5];
            int len = 0, c;
            uint32_t r;
            chain[len++] = pos_items[e];
            for (c = 0; c < 3; c++) {
                const int64_t cnt = ptrs[c][u + 1] - ptrs[c][u];
                if (cnt > 0) {                                       /* choice(list) = list[_randbelow(len)] */
                    const int kb = bit_length_u32((uint32_t)cnt);
                    do { r = mt_u32(s) >> (32 - kb); } while (r >= (uint32_t)cnt);
                    chain[len++] = lists[c][ptrs[c][u] + r];
                }
            }
            for (;;) {
                r = mt_u32(s) >> (32 - kbits);
                if (r >= (uint32_t)n_items) continue;
                if (stamp[r] == u + 1) continue;
                break;
            }
            chain[len++] = (int32_t)r;
            for (c = 0; c + 1 < len; c++) { u_out[n] = u; a_out[n] = chain[c]; b_out[n] = chain[c + 1]; n++; }
        }
    }
    free(stamp);
    return n;
}

/* TBPR.optimization over the chained triplets (same arithmetic as BPR.optimization, and a == b can occur: the last draw
 * may repeat a social item; rows then alias exactly as numpy's in-place row updates do), plus the reference's loss:
 * sum(-log s) and, after EVERY user's updates, regU*sum(P*P) + regI*sum(Q*Q) over the whole tables (TBPR.py:159 sits
 * inside the user loop). */
double orc_tbpr_epoch_f64(double *P, double *Q, int32_t d, int32_t n_users, int32_t n_items, const int32_t *u_idx,
                          const int32_t *a_idx, const int32_t *b_idx, int64_t n, double lr, double regU, double regI) {
    double loss = 0.0;
    int64_t t = 0;
    while (t < n) {
        int64_t e = t, k;
        double sp = 0.0, sq = 0.0;
        while (e < n && u_idx[e] == u_idx[t]) e++;
        loss += orc_bpr_sgd_f64(P, Q, d, u_idx + t, a_idx + t, b_idx + t, e - t, lr, regU, regI);
        for (k = 0; k < (int64_t)n_users * d; k++) sp += P[k] * P[k];
        for (k = 0; k < (int64_t)n_items * d; k++) sq += Q[k] * Q[k];
        loss += regU * sp + regI * sq;
        t = e;
    }
    return loss;
}

/* Same recurrence carried out in fp32 storage/arithmetic (fp64 loss accumulator): the
 * tight comparator for the fp32 HIP kernels (dot-product summation order still differs
 * from the wave butterfly, so agreement is to rounding, not bitwise). */
double orc_bpr_sgd_f32(float *P, float *Q, int32_t d, const int32_t *u_idx,
                       const int32_t *i_idx, const int32_t *j_idx, int64_t n,
                       float lr, float regU, float regI) {
    double loss = 0.0;
    int64_t t;
    int c;
    for (t = 0; t < n; t++) {
        float *pu = P + (int64_t)u_idx[t] * d;
        float *qi = Q + (int64_t)i_idx[t] * d;
        float *qj = Q + (int64_t)j_idx[t] * d;
        float xi = 0.f, xj = 0.f, s, g;
        for (c = 0; c < d; c++) { xi += pu[c] * qi[c]; xj += pu[c] * qj[c]; }
        s = 1.0f / (1.0f + expf(-(xi - xj)));
        g = lr * (1.0f - s);
        for (c = 0; c < d; c++) pu[c] += g * (qi[c] - qj[c]);
        for (c = 0; c < d; c++) qi[c] += g * pu[c];
        for (c = 0; c < d; c++) qj[c] -= g * pu[c];
        for (c = 0; c < d; c++) pu[c] -= (lr * regU) * pu[c];
        for (c = 0; c < d; c++) qi[c] -= (lr * regI) * qi[c];
        for (c = 0; c < d; c++) qj[c] -= (lr * regI) * qj[c];
        /* -log(sigmoid(x)) in the overflow-free form: fp32 sigmoid underflows to 0 below
         * x = -88.7 where the reference's fp64 expression is still finite */
        { double xd = (double)(xi - xj); loss += xd >= 0 ? log1p(exp(-xd)) : -xd + log1p(exp(xd)); }
    }
    return loss;
}

/* model/ranking/BPR.py:40 epoch-end regulariser: (P*P).sum() / (Q*Q).sum(). */
double orc_sumsq_f64(const double *x, int64_t n) {
    double a = 0.0;
    int64_t k;
    for (k = 0; k < n; k++) a += x[k] * x[k];
    return a;
}

/* a-6  BasicMF SGD step: model/rating/BasicMF.py:9-26.
 *   error = rating - P[u].Q[i]; loss += error^2
 *   p,q are VIEWS: P[u] += lr*error*q ; Q[i] += lr*error*p  (p already updated)
 * Rows visited in trainingData order (the caller supplies the current order). */
double orc_mf_sgd_f64(double *P, double *Q, int32_t d, const int32_t *u_idx,
                      const int32_t *i_idx, const double *rating, int64_t n, double lr) {
    double loss = 0.0;
    int64_t t;
    int c;
    for (t = 0; t < n; t++) {
        double *p = P + (int64_t)u_idx[t] * d;
        double *q = Q + (int64_t)i_idx[t] * d;
        double dot = 0.0, err;
        for (c = 0; c < d; c++) dot += p[c] * q[c];
        err = rating[t] - dot;
        loss += err * err;
        for (c = 0; c < d; c++) p[c] += (lr * err) * q[c];
        for (c = 0; c < d; c++) q[c] += (lr * err) * p[c];
    }
    return loss;
}

/* model/rating/PMF.py:9-28 (variant 1) and model/rating/SVD.py:13-35 (variant 2), same walk as BasicMF:
 *   PMF:  error = r - P[u].Q[i];  P[u] += lr*(error*q - regU*p);  Q[i] += lr*(error*p - regI*q)   (p: UPDATED view)
 *   SVD:  error = r - (((P[u].Q[i] + globalMean) + Bi[i]) + Bu[u])   (SVD.py:76-80), same P/Q updates,
 *         Bu[u] += lr*(error - regB*bu);  Bi[i] += lr*(error - regB*bi)    (bu, bi: values read BEFORE the updates)
 *   EE (variant 3, model/rating/EE.py:15-34,80-84): diff = P[u]-Q[i]; error = r - (((globalMean + Bi[i]) + Bu[u]) - diff.diff);
 *         the returned loss also carries regU*diff.diff per rating (EE.py:25);
 *         P[u] -= (lr*(error+regU))*diff;  Q[i] += (lr*(error+regI))*(P[u]-Q[i])  (UPDATED P[u]);  biases as SVD.
 * Returns sum(error^2) (+ the EE term); the epoch-end regularisers are orc_sumsq_f64. */
double orc_mf_sgd_var_f64(int variant, double *P, double *Q, double *Bu, double *Bi, int32_t d,
                          const int32_t *u_idx, const int32_t *i_idx, const double *rating, int64_t n,
                          double lr, double regU, double regI, double regB, double gmean) {
    double loss = 0.0;
    int64_t t;
    int c;
    for (t = 0; t < n; t++) {
        const int32_t u = u_idx[t], i = i_idx[t];
        double *p = P + (int64_t)u * d, *q = Q + (int64_t)i * d;
        double dot = 0.0, pred, err, bu = 0.0, bi = 0.0;
        if (variant == 3) {
            double cu, ci;
            for (c = 0; c < d; c++) { const double df = p[c] - q[c]; dot += df * df; }
            bu = Bu[u]; bi = Bi[i];
            pred = ((gmean + bi) + bu) - dot;
            err = rating[t] - pred;
            loss += err * err;
            loss += regU * dot;
            cu = lr * (err + regU); ci = lr * (err + regI);
            for (c = 0; c < d; c++) p[c] -= cu * (p[c] - q[c]);
            for (c = 0; c < d; c++) q[c] += ci * (p[c] - q[c]);
            Bu[u] += lr * (err - regB * bu); Bi[i] += lr * (err - regB * bi);
            continue;
        }
        for (c = 0; c < d; c++) dot += p[c] * q[c];
        pred = dot;
        if (variant == 2) { bu = Bu[u]; bi = Bi[i]; pred = ((dot + gmean) + bi) + bu; }
        err = rating[t] - pred;
        loss += err * err;
        for (c = 0; c < d; c++) p[c] += lr * (err * q[c] - regU * p[c]);
        for (c = 0; c < d; c++) q[c] += lr * (err * p[c] - regI * q[c]);
        if (variant == 2) { Bu[u] += lr * (err - regB * bu); Bi[i] += lr * (err - regB * bi); }
    }
    return loss;
}

/* model/rating/SVDPlusPlus.py:25-62,70-86, one pass over the ratings in array order (fp64, in place).
 * rated_indptr/rated_items: each user's train items in dict order (data.userRated), w = their count.
 *   pred  = ((sum_j Y[j]) / w) . Q[i]  +  (((P[u].Q[i] + mean) + Bi[i]) + Bu[u])          (:70-86; the sum over ALL w items)
 *   Bu, Bi from their old values; if w > 1, over the user's OTHER items (indexes, dict order):
 *         sum2 = y_0 + y_1 + ... ;  Y[j] = y_j + lr*((err*q)/(w-1) - regY*y_j) ;  Q[i] += ((lr*err)*sum2)/(w-1)
 *   P[u] += lr*(err*q - regU*p)  (q: the UPDATED Q[i], a view) ;  Q[i] += lr*(err*p - regI*q)  (p: the UPDATED P[u])
 * Returns sum(err^2). */
double orc_svdpp_sgd_f64(double *P, double *Q, double *Y, double *Bu, double *Bi, int32_t d,
                         const int64_t *rated_indptr, const int32_t *rated_items,
                         const int32_t *u_idx, const int32_t *i_idx, const double *rating, int64_t n,
                         double lr, double regU, double regI, double regB, double regY, double gmean) {
    double loss = 0.0;
    double *sum = (double *)malloc(sizeof(double) * (size_t)d), *tmp = (double *)malloc(sizeof(double) * (size_t)d);
    int64_t t, k;
    int c;
    for (t = 0; t < n; t++) {
        const int32_t u = u_idx[t], i = i_idx[t];
        double *p = P + (int64_t)u * d, *q = Q + (int64_t)i * d;
        const int64_t b = rated_indptr[u], e = rated_indptr[u + 1];
        const int64_t w = e - b;
        double a = 0.0, dot = 0.0, pred, err, bu = Bu[u], bi = Bi[i];
        if (w > 0) {
            for (c = 0; c < d; c++) sum[c] = 0.0;
            for (k = b; k < e; k++) { const double *y = Y + (int64_t)rated_items[k] * d; for (c = 0; c < d; c++) sum[c] += y[c]; }
            for (c = 0; c < d; c++) a += (sum[c] / (double)w) * q[c];
        }
        for (c = 0; c < d; c++) dot += p[c] * q[c];
        pred = a + (((dot + gmean) + bi) + bu);
        err = rating[t] - pred;
        loss += err * err;
        Bu[u] += lr * (err - regB * bu);
        Bi[i] += lr * (err - regB * bi);
        if (w > 1) {
            int first = 1;
            for (k = b; k < e; k++) {
                const double *y;
                if (rated_items[k] == i) continue;
                y = Y + (int64_t)rated_items[k] * d;
                if (first) { for (c = 0; c < d; c++) sum[c] = y[c]; first = 0; }
                else for (c = 0; c < d; c++) sum[c] += y[c];
            }
            for (c = 0; c < d; c++) tmp[c] = (err * q[c]) / (double)(w - 1);
            for (k = b; k < e; k++) {
                double *y;
                if (rated_items[k] == i) continue;
                y = Y + (int64_t)rated_items[k] * d;
                for (c = 0; c < d; c++) y[c] = y[c] + lr * (tmp[c] - regY * y[c]);
            }
            if (!first) for (c = 0; c < d; c++) q[c] += ((lr * err) * sum[c]) / (double)(w - 1);
        }
        for (c = 0; c < d; c++) p[c] += lr * (err * q[c] - regU * p[c]);
        for (c = 0; c < d; c++) q[c] += lr * (err * p[c] - regI * q[c]);
    }
    free(sum); free(tmp);
    return loss;
}

/* ------------------------------------------------------------------------------------
 * a-15  find_k_largest: util/qmath.py:134-146, including CPython heapq's exact sift
 * order (Lib/heapq.py) because ties are resolved by the heap layout:
 *   heap of the first K (score,iid) tuples (tuple order: score, then iid);
 *   for each later item: if score > heap[0].score (strict): heapreplace;
 *   list.sort(key=score, reverse=True)  -- stable for equal scores.
 */
typedef struct { double s; int32_t id; } orc_pair;
static inline int pair_lt(const orc_pair *a, const orc_pair *b) {
    if (a->s < b->s) return 1;
    if (a->s > b->s) return 0;
    return a->id < b->id;
}
static void heap_siftdown(orc_pair *h, int startpos, int pos) {
    orc_pair newitem = h[pos];
    while (pos > startpos) {
        int parentpos = (pos - 1) >> 1;
        if (pair_lt(&newitem, &h[parentpos])) { h[pos] = h[parentpos]; pos = parentpos; continue; }
        break;
    }
    h[pos] = newitem;
}
static void heap_siftup(orc_pair *h, int n, int pos) {
    int endpos = n, startpos = pos, childpos = 2 * pos + 1;
    orc_pair newitem = h[pos];
    while (childpos < endpos) {
        int rightpos = childpos + 1;
        if (rightpos < endpos && !pair_lt(&h[childpos], &h[rightpos])) childpos = rightpos;
        h[pos] = h[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    h[pos] = newitem;
    heap_siftdown(h, startpos, pos);
}
/* stable insertion sort by score descending == list.sort(key=score, reverse=True) */
static void stable_sort_desc(orc_pair *h, int n) {
    int a, b;
    for (a = 1; a < n; a++) {
        orc_pair x = h[a];
        for (b = a - 1; b >= 0 && h[b].s < x.s; b--) h[b + 1] = h[b];
        h[b + 1] = x;
    }
}
int orc_find_k_largest(int32_t K, const double *cand, int32_t n, int32_t *ids, double *scores) {
    int k = K < n ? K : n, t;
    orc_pair *h = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(k > 0 ? k : 1));
    for (t = 0; t < k; t++) { h[t].s = cand[t]; h[t].id = t; }
    for (t = k / 2 - 1; t >= 0; t--) heap_siftup(h, k, t); /* heapify */
    for (t = k; t < n; t++) {
        if (cand[t] > h[0].s) { h[0].s = cand[t]; h[0].id = t; heap_siftup(h, k, 0); }
    }
    stable_sort_desc(h, k);
    for (t = 0; t < k; t++) { ids[t] = h[t].id; scores[t] = h[t].s; }
    free(h);
    return k;
}
