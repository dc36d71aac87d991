/*
 * qrec_hip.h -- C ABI of libqrec_hip.so, the MI355X (gfx950) implementation of QRec's
 * embedding-training hot path.
 *
 * The reference (Coder-Yu/QRec) is pure Python and has no FFI of its own; its only
 * extension seam is the Recommender template-method API (base/recommender.py:181-212).
 * Each entry point below replaces the body of one reference function; the Python model
 * classes in qrec_amd/ (and the stub a QRec maintainer would add)
 * bind these symbols with ctypes.  Plain C types only: pointers, sizes, scalars.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; qrec_last_error() gives the text
 *     (thread-local).  The reference reports errors with print+exit(-1)
 *     (util/config.py:8-10, base/iterativeRecommender.py:84-86); the Python wrapper maps
 *     error codes to that behaviour.
 *   - pointers named d_* are DEVICE pointers (from qrec_malloc or any HIP allocation,
 *     e.g. a torch tensor's data_ptr()); h_* are HOST pointers, borrowed for the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream unless stated otherwise.
 *   - embedding tables are row-major [rows][ld] with ld >= d, ld a multiple of 4 (fp32) and
 *     the pad columns zero; dtype is QREC_F32 or QREC_F64.
 *   - a handle is not thread-safe; one process drives one device.
 */
#ifndef QREC_HIP_H
#define QREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QREC_OK 0
#define QREC_ERR_INVALID (-1)  /* bad argument */
#define QREC_ERR_HIP (-2)      /* a HIP runtime call failed */
#define QREC_ERR_NAN (-3)      /* loss is NaN/Inf (base/iterativeRecommender.py:84-86) */
#define QREC_ERR_UNSUPPORTED (-4)

#define QREC_F32 0
#define QREC_F64 1
#define QREC_I32 2      /* collectives only */

/* ---- runtime ------------------------------------------------------------------------ */
int qrec_version(void);
const char *qrec_last_error(void);
int qrec_device_count(int *n);
/* Select the device for this process.  Deliberately NOT called from any constructor:
 * QRec builds model objects in the parent and forks per CV fold (QRec.py:76-89). */
int qrec_init(int device);
int qrec_device_info(char *name, int name_len, int *n_cu, int64_t *hbm_bytes, char *arch,
                     int arch_len);
int qrec_malloc(int64_t bytes, void **d_ptr);
int qrec_free(void *d_ptr);
int qrec_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes, void *stream);
int qrec_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes, void *stream);
int qrec_memcpy_d2d(void *d_dst, const void *d_src, int64_t bytes, void *stream);
/* Page-locked host memory and a device-to-host copy into it that only ENQUEUES (qrec_memcpy_d2h returns after the copy, i.e.
 * drains the stream): the host reads h_pinned after an event recorded behind the copy.  For the small read-backs a run-ahead
 * host needs while the stream carries on (the row counts of a sharded exchange plan, qrec_amd/dist.py). */
int qrec_host_alloc(int64_t bytes, void **h_ptr);
int qrec_host_free(void *h_ptr);
int qrec_memcpy_d2h_async(void *h_pinned, const void *d_src, int64_t bytes, void *stream);
int qrec_memset(void *d_dst, int byte, int64_t bytes, void *stream);
int qrec_stream_create(void **stream);
int qrec_stream_destroy(void *stream);
int qrec_stream_sync(void *stream);
int qrec_device_sync(void);
int qrec_event_create(void **ev);
int qrec_event_destroy(void *ev);
int qrec_event_record(void *ev, void *stream);
int qrec_event_sync(void *ev);
/* make `stream` wait (on the device) for work recorded in `ev` */
int qrec_stream_wait_event(void *stream, void *ev);
int qrec_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms);

/* ---- exact (bit-reproducible) sampler, host side ------------------------------------- *
 * Replays CPython's `random` (MT19937) word for word.  `state625` is
 * random.getstate()[1] (624 words + index) and is advanced in place so the Python host
 * can random.setstate() it back and stay in lock-step with the reference.            */

/* model/ranking/BPR.py:28-38: one negative per entry of the PositiveSet CSR, users in id
 * order, items in row order; redraw while the item is a positive of the user. */
int qrec_mt_bpr_sample_epoch(uint32_t *state625, const int64_t *h_pos_indptr,
                             const int32_t *h_pos_indices, int32_t n_users, int32_t n_items,
                             int32_t *h_j_out);
/* random.shuffle of a length-n list (base/iterativeRecommender.py:101,
 * base/deepRecommender.py:30).  h_perm (int64[n]) is permuted in place; NULL only
 * advances the generator. */
int qrec_mt_shuffle(uint32_t *state625, int64_t n, int64_t *h_perm);
/* DataSplit.dataSplit (util/dataSplit.py:9-25): one random() per row of the loaded file, in row order; is_test_out[k]
 * = 1 when random() < ratio.  Same draws as the reference's loop, so the -ap split is the reference's split. */
int qrec_mt_data_split(uint32_t *state625, int64_t n, double ratio, uint8_t *is_test_out);
/* base/deepRecommender.py:41-49 over rows already in shuffled order: one negative per
 * row, redraw while the item is in trainSet_u[user].  rated CSR rows must be sorted. */
int qrec_mt_pairwise_sample_epoch(uint32_t *state625, const int32_t *h_row_user, int64_t n_rows,
                                  const int64_t *h_rated_indptr, const int32_t *h_rated_sorted,
                                  int32_t n_items, int32_t *h_neg_out);

/* random.sample(range(n), k) (model/ranking/SGL.py:118-135: kept edges / dropped nodes of an
 * augmented sub-graph), same generator, same draws as CPython 3.10. */
int qrec_mt_sample_range(uint32_t *state625, int64_t n, int64_t k, int64_t *h_out);

/* ---- throughput sampler, device side -------------------------------------------------- *
 * Same distribution as BPR.py:35-37 (uniform over the items that are not positives of
 * the user, by rejection), counter-based Philox4x32-10 keyed by (seed, epoch, triplet
 * index): reproducible and order-independent, but NOT the CPython stream.
 * The stream is a contract: counter = {t_lo, t_hi, block, epoch_lo}, key = {seed_lo, seed_hi ^ epoch_hi}, block = 0, 1, ...;
 * the block's four words in order, candidate = word >> (32 - bit_length(n_items)), skipped when >= n_items or a positive of
 * d_row_user[t]; first survivor = d_j_out[t]; none in blocks 0 .. 4096: -1.  oracle/qrec_oracle.c orc_philox_bpr_sample
 * restates it (Philox pinned to Random123's known answers) and the kernel is held to it bit for bit.                        */
int qrec_philox_bpr_sample(const int64_t *d_pos_indptr, const int32_t *d_pos_sorted,
                           const int32_t *d_row_user, int64_t n, int32_t n_items, uint64_t seed,
                           uint64_t epoch, int32_t *d_j_out, void *stream);

/* Throughput mode of the pairwise (TF-path) models, base/deepRecommender.py:29-52 on the device: the epoch's row order
 * from qrec_random_permutations (a uniform shuffle, one device sort), the rows gathered into that order here, one
 * negative per row from qrec_philox_bpr_sample with the rated CSR as the exclusion set.  Same distribution as
 * shuffle + choice-with-rejection, not the CPython stream.                                                          */
int qrec_gather_pairs(const int32_t *d_perm, const int32_t *d_u, const int32_t *d_i, int64_t n, int32_t *d_u_out,
                      int32_t *d_i_out, void *stream);

/* ---- BPR SGD: model/ranking/BPR.py:45-53 --------------------------------------------- */

/* Order-exact mode: the n triplets are applied strictly one after another, in array
 * order, each seeing all earlier writes -- the reference's semantics.  One wavefront
 * walks the list (the dependency chain of an epoch is ~n/6 long).
 * *d_loss (double) receives sum(-log(sigmoid(x))) (overwritten).                        */
int qrec_bpr_sgd_ordered(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld,
                         const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int64_t n,
                         double lr, double regU, double regI, double *d_loss, void *stream);

/* Order-exact mode beyond one wavefront.  The negatives are drawn before the first update (sampling never looks at the
 * embeddings), so the epoch's dependence DAG -- triplet t waits for the previous toucher of P[u], Q[i], Q[j], nothing
 * else -- is known up front.  qrec_bpr_exact_schedule (host) list-schedules the triplets, in the reference's order, into
 * steps of at most `width` mutually independent triplets and records for every row where its current value will be
 * (LDS forwarding slot of one of the last two steps, or the table); qrec_bpr_sgd_scheduled executes the steps with one
 * workgroup of `width` wavefronts, one barrier per step.  Every row sees exactly the reference's sequence of updates:
 * results equal qrec_bpr_sgd_ordered's bit for bit (same per-triplet arithmetic), the loss up to its summation order.
 *   h_entries   : int32[n][8] out, step-major: {u, i, j, t, src_P, src_Qi, src_Qj, 0}
 *   h_step_off  : int32[n + 1] capacity out; entries of step s are [h_step_off[s], h_step_off[s+1])
 *   width       : <= qrec_bpr_exact_width(dtype, d) (what the CU's LDS holds), <= QREC_EXACT_MAX_WIDTH
 *   d_xlog      : table-dtype[n + QREC_EXACT_XLOG_PAD] scratch (x per triplet, then the dummy rows idle wavefronts work on),
 *   d_scratch   : double[QREC_EXACT_SCRATCH_WORDS], zero before first use */
#define QREC_EXACT_MAX_WIDTH 16       /* a power of two: a source code is ((dist - 1) * 16 + slot) * 4 + row, or -1 = the table */
#define QREC_EXACT_XLOG_PAD (16 * 256 + 64)
#define QREC_EXACT_SCRATCH_WORDS 130
int qrec_bpr_exact_width(int dtype, int32_t d, int32_t *width);
int qrec_bpr_exact_schedule(const int32_t *h_u, const int32_t *h_i, const int32_t *h_j, int64_t n, int32_t n_users,
                            int32_t n_items, int32_t width, int32_t *h_entries, int32_t *h_step_off, int64_t *n_steps);
int qrec_bpr_sgd_scheduled(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld, const int32_t *d_entries,
                           const int32_t *d_step_off, int64_t n_steps, int32_t width, int64_t n, double lr, double regU,
                           double regI, void *d_xlog, double *d_scratch, double *d_loss, void *stream);
/* Four triplets per wavefront (round 3; rows of 16, 32, 64 or 128 elements): a triplet on the 16 lanes of one DPP row, rows handed
 * on in registers or through the table, nothing in LDS (bpr_exact.hip).  Same order, same values as qrec_bpr_sgd_scheduled up to
 * the last bits (own summation tree and exp: 1e-13 apart in fp64); identical bits across widths.  2.4 x the steps per second.
 *   qrec_bpr_exact_kind       : which kernel serves (dtype, ld, width): 0 = qrec_bpr_exact_schedule + qrec_bpr_sgd_scheduled
 *                               (above), 1 = the three calls below; slots = 4 * ceil(width / 4) for kind 1.
 *                               Env QREC_EXACT_KERNEL=w64 forces 0 (measurement).
 *   qrec_bpr_exact_schedule_reg : host; arguments and entry format of qrec_bpr_exact_schedule, but h_step_off needs 3 n + 2
 *                               entries (steps may stay empty: n_steps <= 3 n).  A user's run stays on one slot and hands P[u]
 *                               on in that group's registers (src_P = -2); every other row is read from the table, so
 *                               touches of a row by different slots are kept >= 3 steps apart.  Entry word 7 = bit 0 (P goes
 *                               on in registers: no table store) | slot << 8 | 0x1000.
 *   qrec_bpr_exact_expand     : d_entries / d_step_off (device copies of that schedule) -> the fixed-width layout the kernel
 *                               reads: d_wide int32[(n_steps + QREC_EXACT_WIDE_PAD) * slots * 8], QREC_EXACT_WIDE_LEAD empty
 *                               steps in front, empty slots u = -1;
 *   qrec_bpr_sgd_scheduled_wide : the epoch; d_xlog / d_scratch / d_loss as above.  The tables are addressed through
 *                               bounds-checked 32-bit offsets: each of P, Q below 4 GiB (else kind 0).                        */
#define QREC_EXACT_WIDE_PAD 16
#define QREC_EXACT_WIDE_LEAD 4
int qrec_bpr_exact_kind(int dtype, int32_t ld, int32_t width, int32_t *kind, int32_t *slots);
int qrec_bpr_exact_schedule_reg(const int32_t *h_u, const int32_t *h_i, const int32_t *h_j, int64_t n, int32_t n_users,
                                int32_t n_items, int32_t width, int32_t *h_entries, int32_t *h_step_off, int64_t *n_steps);
int qrec_bpr_exact_expand(const int32_t *d_entries, const int32_t *d_step_off, int64_t n_steps, int32_t slots, int32_t *d_wide,
                          void *stream);
int qrec_bpr_sgd_scheduled_wide(void *d_P, void *d_Q, int64_t n_users, int64_t n_items, int dtype, int32_t d, int32_t ld,
                                const int32_t *d_wide, int64_t n_steps, int32_t slots, int64_t n, double lr, double regU, double regI,
                                void *d_xlog, double *d_scratch, double *d_loss, void *stream);

/* Throughput mode (fp32): triplets are cut into chunks of `chunk` consecutive entries;
 * one 16-lane (d<=64) / 32-lane (d<=128) group owns a chunk, keeps P[u] in registers
 * along a user run, and applies every row update as an atomic add of the exact
 * per-sample delta, so no update is lost; concurrent chunks read rows that may lag by
 * the in-flight updates (Hogwild).  grid_groups = number of groups in flight (0 = library
 * default: one 256-thread block per CU); with grid_groups==1 the kernel degenerates to the
 * sequential recurrence.  *d_loss (double) is ACCUMULATED into (zero it first).  n_users / n_items = rows of
 * d_P / d_Q: tables below 4 GiB are addressed through a buffer descriptor of exactly that size (a row id past the
 * end is dropped by the hardware bounds check), larger ones through 64-bit addresses (2-5% slower).
 * `variant` selects the memory policy (QREC_HW_*), 0 = library default.                  */
#define QREC_HW_DEFAULT 0
#define QREC_HW_PLAIN_RMW 1     /* plain loads, plain stores (racy read-modify-write)     */
#define QREC_HW_SC1_RMW 2       /* sc1 loads, sc1 write-through stores                    */
#define QREC_HW_ATOMIC 3        /* plain loads, f32 atomic-add deltas                     */
#define QREC_HW_SC1_ATOMIC 4    /* sc1 loads, f32 atomic-add deltas                       */
#define QREC_HW_P_RMW 5         /* item-major only: P[u] by sc1 load + sc1 store (racy), Q[i] / Q[j] by atomic deltas */
#define QREC_HW_PQ_RMW 6        /* item-major only: P[u] and Q[j] by sc1 load + sc1 store (measurements)              */
int qrec_bpr_sgd_hogwild(float *d_P, float *d_Q, int64_t n_users, int64_t n_items, int32_t d, int32_t ld, const int32_t *d_u,
                         const int32_t *d_i, const int32_t *d_j, int64_t n, int32_t chunk,
                         int32_t grid_groups, float lr, float regU, float regI, double *d_loss,
                         int variant, const double *d_driver_state, void *stream);

/* The same Hogwild epoch, ITEM-major: the caller passes the triplets sorted by positive item (stable
 * within an item); Q[i] stays in registers along an item run (flushed as one atomic delta and re-read
 * every `flush_every` triplets), P[u] and Q[j] take the per-sample atomic deltas.  Moves the
 * per-triplet atomics off the hot item rows (Zipf 0.6) onto user rows (Zipf 0.4) and uniform
 * negatives, which the L2 atomic units retire ~25% faster.  Chunks are visited in a golden-ratio
 * stride order so that the chunks of one hot item are spread over the epoch.  With grid_groups == 1
 * and chunk order aside, each row still sees exactly the reference recurrence.
 * `variant`: QREC_HW_DEFAULT / QREC_HW_ATOMIC = the above.  QREC_HW_P_RMW (round 6): P[u] is read with sc1 loads and written back
 * with sc1 write-through stores instead of atomic deltas -- one atomic row update per triplet instead of two (0.63-0.64 of the
 * HBM roofline instead of 0.43-0.46), at the price that a P[u] update landing between another group's load and store of the same
 * row is lost.  The expected share of such updates is the collision density c = groups in flight x sum_u (n_u / n)^2; the host
 * takes this variant only where c <= 0.01 (qrec_amd/engine.py resolve_p_update; measured Recall@20 gaps by c in DESIGN.md s5.1). */
int qrec_bpr_sgd_hogwild_item_major(float *d_P, float *d_Q, int64_t n_users, int64_t n_items, int32_t d, int32_t ld,
                                    const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int64_t n, int32_t chunk,
                                    int32_t grid_groups, int32_t flush_every, float lr, float regU, float regI,
                                    double *d_loss, int variant, const double *d_driver_state, void *stream);
/* Device-resident epoch close of the numpy-path models: model/ranking/BPR.py:40 (loss += regU*sum(P*P) +
 * regI*sum(Q*Q)) followed by isConverged / updateLearningRate (base/iterativeRecommender.py:56-63,88-104),
 * so that a run can be enqueued epoch after epoch without a host round trip.
 *   d_stats : double[QREC_STATS_WORDS], zero before the first epoch: [0] the epoch's sum(-log sigma) (the SGD
 *             kernels accumulate into it; cleared again by the call), [1] sum P*P, [2] sum Q*Q (left for the
 *             caller), [3] ticket counter, then the per-block partial sums.
 *   d_state : double[QREC_DRV_WORDS], the bold-driver state below; initialise LR, everything else 0.
 *   d_log   : optional double[log_capacity][QREC_DRV_LOG_WORDS] = {loss, lr used, sum(-log sigma),
 *             lastLoss - loss, sum P*P, sum Q*Q, -, -} per epoch.
 * A SGD entry point given d_driver_state takes its learning rate from d_state[QREC_DRV_LR] and becomes a
 * no-op once CONVERGED or FAILED is set (tol = 1e-3 in the reference; 0 never converges; max_lr <= 0 = no cap). */
#define QREC_DRV_LR 0         /* learning rate of the NEXT epoch (lRate)             */
#define QREC_DRV_LAST_LOSS 1  /* lastLoss                                            */
#define QREC_DRV_EPOCHS 2     /* epochs closed so far                                */
#define QREC_DRV_CONVERGED 3  /* 1 once |lastLoss - loss| < tol                      */
#define QREC_DRV_FAILED 4     /* 1 once the loss is NaN/Inf (the reference exits)    */
#define QREC_DRV_WORDS 8
#define QREC_DRV_LOG_WORDS 8
#define QREC_STATS_MAX_BLOCKS 256
#define QREC_STATS_PARTIALS 8
#define QREC_STATS_WORDS (QREC_STATS_PARTIALS + 2 * QREC_STATS_MAX_BLOCKS)
int qrec_epoch_close(const void *d_P, int64_t p_rows, const void *d_Q, int64_t q_rows, int dtype, int32_t ld,
                     double *d_stats, double *d_state, double regU, double regI, double max_lr, double tol,
                     double *d_log, int64_t log_capacity, void *stream);
/* The two halves, for runs whose loss terms are summed over ranks in between (one process per GPU): qrec_epoch_sums
 * leaves sum P*P / sum Q*Q in d_stats[1..2] (d_state may be NULL; a converged/failed state makes it a no-op), the
 * caller all-reduces d_stats[0..1] (sum(-log sigma) and sum P*P add up over user shards, Q is replicated), and
 * qrec_epoch_decide takes the reference's decision from d_stats[0..2] exactly as qrec_epoch_close does. */
int qrec_epoch_sums(const void *d_P, int64_t p_rows, const void *d_Q, int64_t q_rows, int dtype, int32_t ld,
                    double *d_stats, const double *d_state, void *stream);
int qrec_epoch_decide(double *d_stats, double *d_state, double regU, double regI, double max_lr, double tol,
                      double *d_log, int64_t log_capacity, void *stream);
/* one table's (deterministic) sum of squares into d_stats[slot], slot 1 = sum P*P, 2 = sum Q*Q; the other slot is left alone */
int qrec_epoch_sum_table(const void *d_X, int64_t rows, int dtype, int32_t ld, double *d_stats, int slot,
                         const double *d_state, void *stream);

/* Rating-prediction MF family, order-exact: variant 0 = model/rating/BasicMF.py:9-26 (config #1),
 * 1 = model/rating/PMF.py:9-28 (regU, regI), 2 = model/rating/SVD.py:13-35 (biases d_Bu/d_Bi of the
 * tables' dtype, regB, global mean), 3 = model/rating/EE.py:15-34 (Euclidean embedding; *d_loss then also
 * carries the per-rating regU*|P[u]-Q[i]|^2 term of EE.py:25).  Rows are visited in array order (the caller passes the current
 * trainingData order); *d_loss receives sum(error^2).                                          */
int qrec_mf_sgd_ordered(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld,
                        const int32_t *d_u, const int32_t *d_i, const double *d_rating, int64_t n,
                        double lr, double *d_loss, int variant, double regU, double regI, void *d_Bu,
                        void *d_Bi, double regB, double global_mean, void *stream);

/* model/rating/SVDPlusPlus.py:25-62,70-86, order-exact: the implicit-feedback table d_Y [n_items][ld] next to P, Q and
 * the biases; d_rated_indptr/d_rated_items = each user's train items in data.userRated order (CSR over all users).
 * *d_loss receives sum(error^2); the epoch-end regularisers (:64-65) are qrec_sumsq.                        */
int qrec_svdpp_sgd_ordered(void *d_P, void *d_Q, void *d_Y, void *d_Bu, void *d_Bi, int dtype, int32_t d, int32_t ld,
                           const int64_t *d_rated_indptr, const int32_t *d_rated_items, const int32_t *d_u,
                           const int32_t *d_i, const double *d_rating, int64_t n, double lr, double regU, double regI,
                           double regB, double regY, double global_mean, double *d_loss, void *stream);

/* sum(x*x) over rows x d of a table (epoch-end regulariser, BPR.py:40); *d_out
 * (double) is overwritten. */
int qrec_sumsq(const void *d_x, int dtype, int64_t rows, int32_t d, int32_t ld, double *d_out,
               void *stream);

/* ---- graph recommenders (LightGCN family), TF-1.14 op semantics ----------------------- */

/* tf.sparse_tensor_dense_matmul(norm_adj, X) (model/ranking/LightGCN.py:17, NGCF.py:28,
 * SimGCL.py:25,33):  Y = A X  [+ addend_scale * addend]  and, if d_accum, accum += Y (the
 * running layer sum of LightGCN.py:19).  A is CSR (d_indices/d_values) cut by the host into
 * SEGMENTS of at most a few hundred non-zeros so that heavy rows do not serialise:
 * segment s covers non-zeros [seg_beg, seg_beg+seg_len) of row seg_row; seg_slot < 0 means
 * "the whole row, write Y directly", otherwise the partial sum goes to scratch slot
 * seg_slot of d_partial and long row k (long_row[k]) is the in-order sum of its
 * long_count[k] slots starting at long_first[k].  X, Y, addend, accum: [rows][ld] fp32,
 * ld in {32,64,128,256}.  Deterministic (no float atomics).  d_x_row_mask (may be NULL):
 * bitmap, bit c set iff row c of X is non-zero; clear rows are skipped (bit-identical result,
 * less traffic) -- used for the first backward SpMM, whose operand is the sparse batch gradient.
 * d_y_row_mask (may be NULL): bitmap of the OUTPUT rows that are wanted; the others are neither computed nor
 * written (Y, accum keep what they held) -- the last propagation layer of a training step is only read at the
 * batch's rows (embedding_lookup, LightGCN.py:22-24; qrec_mark_batch_rows builds the bitmap).
 * d_accum_init (may be NULL; needs d_accum): accum = accum_init + Y instead of accum += Y -- the first layer starts the
 * layer sum from the input table (LightGCN.py:15-19: all_embeddings = [ego_embeddings]), no copy before it.
 * d_addend_row_mask (may be NULL; needs d_addend): bitmap of the rows at which the addend is defined; elsewhere it counts as
 * zero and is not read -- the addend of the backward recurrences is the batch gradient, non-zero at the batch's rows only. */
int qrec_spmm_csr(const int32_t *d_seg_row, const int64_t *d_seg_beg, const int32_t *d_seg_len,
                  const int32_t *d_seg_slot, int64_t n_segs, const int32_t *d_long_row,
                  const int32_t *d_long_first, const int32_t *d_long_count, int32_t n_long,
                  const int32_t *d_indices, const float *d_values, const float *d_X, float *d_Y,
                  float *d_partial, int32_t ld, const float *d_addend, float addend_scale, float *d_accum,
                  const float *d_accum_init, const uint32_t *d_x_row_mask, const uint32_t *d_y_row_mask,
                  const uint32_t *d_addend_row_mask, void *stream);
/* d_row_mask |= bits of rows u[b], n_users+i[b], n_users+j[b] (bitmap over the joint [U;V] row space; clear it first) */
int qrec_mark_batch_rows(const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int32_t B, int32_t n_users,
                         uint32_t *d_row_mask, void *stream);

/* embedding_lookup x3 + util/loss.py:3-6 bpr_loss + the batch l2 term and all their gradients
 * (LightGCN.py:22-30): rows are S[row]/div (div = n_layers+1 folds the layer mean in), users
 * at rows [0,n_users), items at n_users+id.  dE (pre-zeroed, [n_rows][ld]) receives the
 * scatter-added row gradients; *d_loss (double) is ACCUMULATED into.  d_row_mask (may be NULL;
 * pre-zeroed, (n_rows+31)/32 words): bit r is set for every row of dE that receives a gradient.
 * d_ordered_ws (may be NULL): NULL = the row gradients are added with float atomics (throughput mode: right to fp32
 * rounding, summation order differs from launch to launch); a workspace of qrec_ordered_scatter_workspace_bytes(3 B, ld)
 * bytes = parity mode: every row's gradient is the sum of its lookups' slots in batch order, one lookup (u, i, j) at a
 * time -- the order of the reference's CPU scatter (UnsortedSegmentSum per IndexedSlices) -- same bits on every launch. */
int qrec_bpr_batch_loss_grad(const float *d_S, float div, int32_t n_users, int64_t n_rows, int32_t ld,
                             const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int32_t B, float eps,
                             float reg, float *d_dE, double *d_loss, uint32_t *d_row_mask, void *d_ordered_ws,
                             int64_t ordered_ws_bytes, void *stream);

/* Ordered scatter-add of rows (csrc/ordered.hip), the deterministic form of the gradient of tf.nn.embedding_lookup
 * (LightGCN.py:22-24, BUIR.py:88-95, SEPT.py:239; TF's UnsortedSegmentSum walks the slots in order on the CPU):
 *   d_out[d_dst_rows[s]] += sum of d_src[s][.] over the slots s with that destination, ascending s, class by class
 *   (class = s / class_size, the classes' sums added in class order; class_size 0 = one class); rows < 0 are skipped.
 * One stable rocPRIM radix sort of (row, slot) + one walk per run; fp32 adds in exactly that order, no contraction:
 * the result equals numpy's np.add.at(out, rows, src) bit for bit when out starts at zero.
 * Workspace: qrec_ordered_scatter_workspace_bytes(n_slots, ld) (also the size the d_ordered_ws arguments of
 * qrec_bpr_batch_loss_grad / qrec_buir_batch_loss_grad / qrec_sept_ssl_loss_grad take: 3 B, 2 B, n k slots). */
int qrec_ordered_scatter_workspace_bytes(int64_t n_slots, int32_t ld, int64_t *bytes);
int qrec_scatter_add_rows_ordered(const float *d_src, const int32_t *d_dst_rows, int64_t n_slots, int32_t ld, int64_t class_size,
                                  float *d_out, void *d_workspace, int64_t workspace_bytes, void *stream);

/* tf.train.AdamOptimizer dense update (LightGCN.py:31-32) in TF 1.14's ApplyAdam form, fp32,
 * with g = grad_scale * d_grad + grad_l2 * theta (grad_l2 = reg folds in d/dtheta of
 * reg*tf.nn.l2_loss(theta), model/ranking/BPR.py:83):  m += (g-m)(1-beta1); v += (g*g-v)(1-beta2);
 * theta -= (m*alpha)/(sqrt(v)+eps).  alpha = lr*sqrt(1-beta2^t)/(1-beta1^t) is supplied by
 * the host, which keeps the fp32 beta powers like TF does.                               */
int qrec_adam_step(float *d_theta, float *d_m, float *d_v, const float *d_grad, int64_t n_elems, float grad_scale,
                   float grad_l2, float alpha, float beta1, float beta2, float eps, void *stream);

/* ---- SimGCL (model/ranking/SimGCL.py) ---------------------------------------------------- */

/* perturbed_LightGCN_encoder's noise step (SimGCL.py:33-35) on one layer output: with x = d_src (or d_emb itself
 * when d_src is NULL: in place)  emb = x + sign(x) * l2_normalize(noise, axis=1) * eps, then (if d_accum) accum += emb.
 * The out-of-place form lets the two perturbed encoders share the first product A E with the clean one.
 * d_noise = [n_rows][ld] U[0,1) numbers, or NULL to draw them on the device with
 * Philox4x32-10(key=seed, counter={row, lane, stream_id}) -- same distribution as
 * tf.random.uniform, not TF's stream.
 * Injected noise may carry the sign to use in place of sign(x) (parity tests that follow a recorded run of the
 * reference: sign() is discontinuous at 0, so an entry of x within rounding of zero can take the other sign here
 * than it did there): a value v in [0, 1) is plain noise; +-(2 + u), u in [0, 1): noise u with the sign forced to
 * +-1; 4 + u: noise u with the sign forced to 0.  The l2_normalize of the row uses u.          */
int qrec_perturb_rows(float *d_emb, const float *d_src, int64_t n_rows, int32_t d, int32_t ld, float eps,
                      const float *d_noise, uint64_t seed, uint64_t stream_id, float *d_accum, const int32_t *d_row_ids,
                      const int32_t *d_n_row_ids, int32_t max_row_ids, int64_t philox_row0, void *stream);
/* The FIRST layer of SimGCL's three encoders in one pass (they share the product A E, SimGCL.py:23-36): emb_v = the
 * perturbed d_src (v = 1, 2: own noise / Philox stream each), and the three layer sums START here -- sum_v = emb_v,
 * src_sum = d_src (assigned, not accumulated: no zero-fill of the sums).  Row subset as for the NGCF calls below
 * (d_row_ids may be NULL): the last layer of a training step is read at the batch's rows only.  philox_row0 (both
 * calls): row r of these tables is row philox_row0 + r of the whole model (a rank's block of row-partitioned tables);
 * the Philox draws are keyed by that row. */
int qrec_perturb_two_views(const float *d_src, float *d_emb1, float *d_emb2, int64_t n_rows, int32_t d, int32_t ld, float eps,
                           const float *d_noise1, const float *d_noise2, uint64_t seed, uint64_t stream_id1, uint64_t stream_id2,
                           float *d_sum1, float *d_sum2, float *d_src_sum, const int32_t *d_row_ids, const int32_t *d_n_row_ids,
                           int32_t max_row_ids, int64_t philox_row0, void *stream);

/* One side (users or items) of SimGCL.calc_cl_loss (SimGCL.py:60-90) with its gradients.
 * x1 = S1[rows]/div, x2 = S2[rows]/div (the two perturbed views' rows of the batch's UNIQUE
 * nodes, n <= 16384, ids distinct); z = l2_normalize(x); loss = -sum log(exp(z1.z2/tau) /
 * sum_cols exp(z1 z2^T/tau)).  *d_loss (double) += loss (unscaled); d_out[rows] += cl_rate *
 * (dloss/dx1 + dloss/dx2), or, when d_out2 is given (SGL: the two views have different backward
 * operators, model/ranking/SGL.py:192-217), d_out += cl_rate*dloss/dx1 and d_out2 += cl_rate*dloss/dx2.
 * The n x n block runs on the f32 MFMA.                                                   */
int qrec_info_nce_workspace_bytes(int32_t n, int32_t ld, int64_t *bytes);
int qrec_info_nce_loss_grad(const float *d_S1, const float *d_S2, float div, const int32_t *d_rows, int32_t n,
                            int32_t ld, float tau, float cl_rate, void *d_workspace, float *d_out, float *d_out2,
                            double *d_loss, void *stream);

/* ---- NGCF dense layers (model/ranking/NGCF.py:27-42) ------------------------------------ *
 * Tables [rows][ld] fp32, ld in {32,64,128}; weights zero-padded to [ld][ld].             */

/* ROW SUBSETS.  The last layer of a TRAINING step is only read at the batch's rows (embedding_lookup, NGCF.py:44-46)
 * and its backward is zero everywhere else, so the three calls below take an optional ascending row list
 * d_row_ids[0 .. *d_n_row_ids) (device memory, built by qrec_compact_marked_rows; max_row_ids = the host's upper
 * bound on the count, used for the launch geometry): only the listed rows are read, computed and written, and the
 * weight gradients sum over the listed rows only.  d_row_ids == NULL: all n_rows rows.  Needs ld <= 64 for the two
 * MFMA calls.  Rows are addressed by their table index either way (no compacted copies).                    */

/* rows[0 .. *d_count) = the rows whose bit is set in a row bitmap (qrec_mark_batch_rows), ascending; at most
 * `capacity` rows are written and counted. */
int qrec_compact_marked_rows(const uint32_t *d_row_mask, int64_t n_rows, int32_t *d_rows, int32_t *d_count, int32_t capacity,
                             void *stream);

/* The row bitmap of a batch (as memset + qrec_mark_batch_rows would leave it) AND its ascending row list in one launch
 * (tables up to 2^20 rows: the bitmap is built in LDS; larger tables take the three separate steps).  d_rows / d_count may
 * both be NULL (bitmap only).  d_zero8[0 .. n_zero8) (n_zero8 <= 64, may be 0): doubles cleared by the same launch -- the
 * step's loss accumulators, which would otherwise cost a memset launch of their own. */
int qrec_mark_compact_batch_rows(const int32_t *d_u, const int32_t *d_i, const int32_t *d_j, int32_t B, int32_t n_users,
                                 int64_t n_rows, uint32_t *d_row_mask, int32_t *d_rows, int32_t *d_count, int32_t capacity,
                                 double *d_zero8, int32_t n_zero8, void *stream);

/* tf.unique of EVERY batch of an epoch's id stream (SimGCL.py:61-64), one launch: for batch b = ids[b*batch .. ) the
 * distinct ids in [0, id_range) (id_range <= 2^20), ascending, + out_offset, go to d_rows[b*batch ..) and their number to
 * d_counts[b].  Ascending instead of tf.unique's first-appearance order: the consumers (InfoNCE sums) are order-free. */
int qrec_unique_per_batch(const int32_t *d_ids, int64_t n, int32_t batch, int32_t id_range, int32_t out_offset, int32_t *d_rows,
                          int32_t *d_counts, void *stream);

/* pre = (side + E) W1 + (E * side) W2   (NGCF.py:29-31; side = A E from qrec_spmm_csr). f32 MFMA. */
int qrec_ngcf_dense_fwd(const float *d_E, const float *d_side, const float *d_W1, const float *d_W2, int64_t n_rows,
                        int32_t ld, float *d_pre, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                        void *stream);
/* In place on d_pre_gate (in: pre, out: backward gate): nxt = dropout(leaky_relu(pre, 0.2), keep)
 * (NGCF.py:32-38; keep = 1 for the inference graph; d_mask = injected 0/1 keep decisions or NULL
 * for device Philox draws), z = l2_normalize(nxt) written to columns [col_off, col_off+d) of the
 * wide table (the concat of NGCF.py:42), 1/|nxt| to d_inv_norm.  philox_row0: row r of these tables is row
 * philox_row0 + r of the whole model (a rank's block of row-partitioned tables): the Philox draws are keyed by that row,
 * so a partitioned run drops the same entries as the single-GPU run.                      */
int qrec_ngcf_activate(float *d_pre_gate, int64_t n_rows, int32_t d, int32_t ld, float keep, const float *d_mask,
                       uint64_t seed, uint64_t stream_id, float *d_next, float *d_wide, int32_t wide_ld,
                       int32_t col_off, float *d_inv_norm, const int32_t *d_row_ids, const int32_t *d_n_row_ids,
                       int32_t max_row_ids, int64_t philox_row0, void *stream);
/* Backward of one layer: dnxt = dE_next (may be NULL) + normalize_bwd(dWide block); dpre = dnxt*gate;
 * dside = dpre W1^T + (dpre W2^T)*E ; dE = dpre W1^T + (dpre W2^T)*side (caller adds A^T dside);
 * gW1 = (side+E)^T dpre, gW2 = (E*side)^T dpre (deterministic two-stage reduction over nodes).
 * With a row subset d_dpre / d_dside / d_dE are written at the listed rows only (the caller zero-fills what it reads
 * elsewhere); d_partial must hold qrec_ngcf_wgrad_partial_bytes(n_rows, ld) bytes either way.
 * d_wide_row_mask (may be NULL): row bitmap of the rows at which d_dWide is defined (the batch's rows, cleared with
 * qrec_zero_rows before the loss scatter instead of zero-filling the whole wide table); elsewhere its block counts as 0. */
int qrec_ngcf_layer_bwd(const float *d_dE_next, const float *d_dWide, const float *d_wide, int32_t wide_ld,
                        int32_t col_off, const float *d_inv_norm, const float *d_gate, const float *d_E,
                        const float *d_side, const float *d_W1, const float *d_W2, int64_t n_rows, int32_t d,
                        int32_t ld, float *d_dpre, float *d_dside, float *d_dE, float *d_partial, float *d_gW1,
                        float *d_gW2, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                        const uint32_t *d_wide_row_mask, void *stream);
int qrec_ngcf_wgrad_partial_bytes(int64_t n_rows, int32_t ld, int64_t *bytes);
/* dst[row][c] (=|+=) src[row][src_col_off + c], c < d : moves the ego block in and out of the wide table; with a row
 * subset (see above) only the listed rows are touched. */
int qrec_copy_cols(float *d_dst, int32_t dst_ld, const float *d_src, int32_t src_ld, int32_t src_col_off, int64_t n_rows,
                   int32_t d, int32_t accumulate, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids,
                   void *stream);
/* X[row][0 .. ld) = 0 for the listed rows (a row subset as above): clears a gradient table where the next scatter lands */
int qrec_zero_rows(float *d_X, int32_t ld, const int32_t *d_row_ids, const int32_t *d_n_row_ids, int32_t max_row_ids, void *stream);

/* ---- per-epoch graph augmentation on the device (SGL.py:113-155, BUIR.py:41-65; throughput mode) ----------------------------
 * A sub-graph of the training graph as a VALUE array over the full graph's CSR structure (csrc/augment.hip): entry e of the joint
 * adjacency gets fl32(fl32(d'_r a'_e) d'_c) with a'_e = the number of kept training rows that map to it and d' = deg'^-1/2 of the
 * kept edges (0 for isolated nodes), dropped entries 0 -- the reference's re-normalised sub-adjacency, laid out for the SpMM plan
 * that already exists.  d_u / d_i: the training rows (user id, item id), d_pos_ui / d_pos_iu: the CSR positions of a row's two
 * entries (u, U + i) and (U + i, u); d_row_of_nnz / d_indices: row and column of every CSR entry; d_dinv_table[k] = float32
 * power(k, -0.5) for k = 0 .. max_deg (inf -> 0; formed by the host with numpy, so the values carry the reference's bits).
 * The draw is the caller's (qrec_random_permutations): d_keep_rows = n_keep kept training rows (edge dropout, random walk) OR
 * d_drop_users / d_drop_items = dropped node ids (node dropout; d_flags: n_users + n_items bytes of scratch) OR neither (the full
 * graph).  Scratch: d_cnt int32[nnz], d_deg int32[n_users + n_items].  Integer atomics only: the result is deterministic. */
int qrec_subgraph_values(const int32_t *d_u, const int32_t *d_i, const int32_t *d_pos_ui, const int32_t *d_pos_iu, int64_t n_edges,
                         int32_t n_users, int32_t n_items, const int32_t *d_keep_rows, int64_t n_keep, const int32_t *d_drop_users,
                         int64_t n_drop_users, const int32_t *d_drop_items, int64_t n_drop_items, const int32_t *d_row_of_nnz,
                         const int32_t *d_indices, int64_t nnz, const float *d_dinv_table, int32_t max_deg, int32_t *d_cnt,
                         int32_t *d_deg, uint8_t *d_flags, float *d_values, void *stream);

/* ---- BUIR (model/ranking/BUIR.py) ------------------------------------------------------------------ *
 * The propagation of both encoders is qrec_spmm_csr on the epoch's two sub-graphs.  What the model adds:      */

/* Per batch element b (u = d_u[b], i = n_users + d_i[b]), with x = S_online[row]/div, t = S_target[row]/div
 * (div = n_layers + 1 folds the layer mean in): q = tanh(x W + bias) (BUIR.py:105-107),
 * *d_loss += [(1 - cos(q_u, t_i)) + (1 - cos(q_i, t_u))] / 2 (:127-129, tf.math.l2_normalize eps 1e-12), the gradient
 * w.r.t. the online mean rows is scatter-added into d_dS (un-divided: the caller folds 1/div into its optimizer
 * step), and the (x, dpre) pairs go to d_X / d_dPre rows b (user side) and B + b (item side) for qrec_buir_wgrad.
 * d_W [ld][ld], d_bias [ld], zero-padded; tables [rows][ld], ld in {32, 64, 128}.
 * d_ordered_ws (nullable, qrec_ordered_scatter_workspace_bytes(2 B, ld)): parity mode -- the scatter into d_dS adds each
 * row's slots in batch order instead of with float atomics (see qrec_bpr_batch_loss_grad).                */
int qrec_buir_batch_loss_grad(const float *d_S_online, const float *d_S_target, float div, int32_t n_users, int32_t ld,
                              const float *d_W, const float *d_bias, const int32_t *d_u, const int32_t *d_i, int32_t B,
                              float *d_dS, float *d_X, float *d_dPre, double *d_loss, void *d_ordered_ws, int64_t ordered_ws_bytes,
                              void *stream);
/* d_gW = X^T dPre, d_gb = column sums of dPre over n_rows (= 2B) pairs; deterministic two-stage sum through
 * d_scratch (qrec_buir_wgrad_scratch_bytes).                                                              */
int qrec_buir_wgrad_scratch_bytes(int32_t ld, int64_t *bytes);
int qrec_buir_wgrad(const float *d_X, const float *d_dPre, int32_t n_rows, int32_t ld, float *d_scratch, float *d_gW,
                    float *d_gb, void *stream);
/* target = target*tau + online*(1 - tau) (BUIR.py:120-123, run after every optimizer step :159)           */
int qrec_ema_update(float *d_target, const float *d_online, float tau, int64_t n_elems, void *stream);

/* ---- full-rank evaluation: base/recommender.py:143-150 + util/qmath.py:134-146 -------- *
 * For each of the n_batch_users users (ids into the user table): scores = V . U[user]
 * (MFMA), scores of the user's rated train items set to 0 (rated CSR over ALL users, may be
 * NULL), then the reference's find_k_largest: min-heap of (score,id) seeded with the first
 * N items, strict '>' replacement, stable descending sort -- ties included, so ids are
 * bit-identical to the reference's.  Outputs [n_batch_users][N] (ids -1 padded when
 * n_items < N).  Tables are [rows][ld] with ld a multiple of 32 floats / 16 doubles and columns [d, ld) zero.
 * N <= 100 as in base/recommender.py:132-134.  Two routes, same lists:
 *   fused (fp32, ld <= 128, N <= 63, n_items >= 16,384): a per-user threshold, then a pass that keeps only the items
 *     whose score reaches it -- no users x items block is ever written.  For ld = 32 / 64 / 128 both passes run on bf16
 *     copies of the tables (v_mfma_f32_32x32x16_bf16) against thresholds lowered by a proven bound on the rounding, and the
 *     few dozen survivors per user are re-scored by the block route's own fp32 MFMA sequence, so ids and scores are
 *     bit-identical to the block route's (env QREC_EVAL_F32_FILTER: both passes in fp32, as for other ld).  Users whose
 *     N+1 best scores are not pairwise distinct (the only case in which the heap's history shows) get the exact
 *     sequential walk, a few hundred at a time.  The rated CSR must then have ASCENDING item ids inside a row.  The call
 *     synchronises `stream` once (it reads back how many such users there are).
 *   block (everything else; forced by env QREC_EVAL_BLOCK_PATH): scores into a transposed users x items block in
 *     d_scratch, mask, sliced top-N with exact fallbacks.
 * Size d_scratch with qrec_score_topk_scratch_bytes (same dtype / sizes / ld / N).                                  */
int qrec_score_topk_scratch_bytes(int dtype, int32_t n_items, int32_t n_batch_users, int32_t ld, int32_t N, int64_t *bytes);
int qrec_score_topk(const void *d_U, const void *d_V, int dtype, int32_t d, int32_t ld, int32_t n_items,
                    const int32_t *d_user_ids, int32_t n_batch_users, const int64_t *d_rated_indptr,
                    const int32_t *d_rated_items, int32_t N, void *d_scratch, int32_t *d_ids_out,
                    void *d_scores_out, void *stream);

/* Measure.hits (util/measure.py:15-21) + the DCG sum of Measure.NDCG (util/measure.py:70-82) over the lists of a
 * previous qrec_score_topk, still on the device: for batch row b (user d_user_ids[b]) and its first n_cut ids,
 * d_hits_out[b] = how many are among the user's test items (CSR over ALL users, item ids ascending inside a
 * row, items unknown to the training set left out), d_dcg_out[b] = sum of d_discount[pos] over the hit
 * positions, added in rank order.  The caller passes discount[pos] = 1/log(pos+2) as ITS doubles, so the
 * per-user values -- and the Precision/Recall/NDCG strings built from them -- equal the reference's bit for bit. */
int qrec_rank_hits(const int32_t *d_ids, int32_t n_batch_users, int32_t row_stride, int32_t n_cut,
                   const int32_t *d_user_ids, const int64_t *d_test_indptr, const int32_t *d_test_items,
                   const double *d_discount, int32_t *d_hits_out, double *d_dcg_out, void *stream);

/* ---- the on-disk rating format: util/io.py:31-76 (FileIO.loadDataSet) --------------------------- *
 * "user item rating" text, one record per line, fields split on every single character of `delims`
 * (NULL = the reference's " ,\t"), columns picked by index (col_rating < 0: no rating column, rating = 1),
 * skip_header drops the first line, binarize drops records rated below `threshold` and sets the rest to 1.
 * The result holds dense ids in first-appearance order, the ratings and the '\n'-joined name tables
 * (which: 0 = users, 1 = items).  Host code, no GPU.  QREC_ERR_UNSUPPORTED = the file needs CPython's own
 * parsing rules (non-ASCII text, exotic float literals, short records): the caller then runs the Python path. */
typedef struct qrec_ratings qrec_ratings;
int qrec_ratings_load(const char *path, const char *delims, int32_t col_user, int32_t col_item, int32_t col_rating,
                      int32_t skip_header, int32_t binarize, double threshold, qrec_ratings **out);
int64_t qrec_ratings_rows(const qrec_ratings *h);
int32_t qrec_ratings_count(const qrec_ratings *h, int32_t which);
int64_t qrec_ratings_names_bytes(const qrec_ratings *h, int32_t which);
int qrec_ratings_copy(const qrec_ratings *h, int32_t *user_out, int32_t *item_out, double *rating_out);
int qrec_ratings_names(const qrec_ratings *h, int32_t which, char *out);
void qrec_ratings_free(qrec_ratings *h);

/* ---- SEPT (model/ranking/SEPT.py): per-layer normalised views, pseudo labels, several-positives contrastive loss ----
 * qrec_l2norm_rows_accum: inv[r] = rsqrt(max(|X[r]|^2, 1e-12)), S[r] += X[r] * inv[r]  -- one propagated layer entering a
 *   view's sum as tf.math.l2_normalize(x, axis=1) (SEPT.py:144-160).  qrec_l2norm_rows_bwd: its gradient w.r.t. X[r],
 *   out[r] = (dS[r] - z (z . dS[r])) * inv[r], z = X[r] * inv[r] (autodiff of the same lines).  qrec_scale_copy:
 *   dst = alpha * src (the Variable / 2 of SEPT.py:129-130).  ld in {32, 64, 128, 256} floats, rows zero-padded. */
int qrec_l2norm_rows_accum(const float *d_X, int64_t n_rows, int32_t ld, float *d_S, float *d_inv, void *stream);
int qrec_l2norm_rows_bwd(const float *d_X, const float *d_inv, const float *d_dS, int64_t n_rows, int32_t ld, float *d_out,
                         void *stream);
int qrec_scale_copy(float *d_dst, const float *d_src, int64_t n_elems, float alpha, void *stream);
/* qrec_sept_ssl_loss_grad: label_prediction + generate_pesudo_labels + neighbor_discrimination (SEPT.py:214-262) on the
 *   n distinct users of a batch (d_rows, tf.unique order).  z_v = l2_normalize(S_v[rows]) for the friend / sharing /
 *   preference views, a = l2_normalize(S_aug[rows]); prob_v = softmax(z_v a^T); labels_v = top_k of the OTHER two
 *   encoders' averaged prob (k = ins_cnt, equal scores in index order);
 *   loss += -sum_i log( sum_{l in labels_v[i]} e^{z_v[i].a[l]/0.1} / sum_j e^{z_v[i].a[j]/0.1} ) over the three encoders
 *   (UNSCALED, added to *d_loss); dS_v[rows[i]] += ss_rate * d loss / d S_v[rows[i]] for the four tables.
 *   d_labels (nullable): int32 [3][n][k] pseudo labels (positions in d_rows) for inspection.
 *   d_ordered_ws (nullable, qrec_ordered_scatter_workspace_bytes(n k, ld)): parity mode -- the positives' part of the augmented
 *   view's gradient (rows shared by several users' label sets) is added per row in (user, label) order instead of with float atomics. */
int qrec_sept_ssl_workspace_bytes(int32_t n, int32_t ld, int32_t k, int64_t *bytes);
int qrec_sept_ssl_loss_grad(const float *d_S_friend, const float *d_S_sharing, const float *d_S_pref, const float *d_S_aug,
                            const int32_t *d_rows, int32_t n, int32_t ld, int32_t k, float ss_rate, void *d_workspace,
                            float *d_dS_friend, float *d_dS_sharing, float *d_dS_pref, float *d_dS_aug, double *d_loss,
                            int32_t *d_labels, void *d_ordered_ws, int64_t ordered_ws_bytes, void *stream);

/* ---- TBPR: model/ranking/TBPR.py (numpy path) -----------------------------------------------------------------
 * qrec_mt_tbpr_sample_epoch (host): the sampling loop TBPR.py:131-158 on the CPython MT19937 stream -- per user and
 *   positive item the chain [i, choice(joint)?, choice(weak)?, choice(strong)?, k] (k: random item, redrawn while
 *   positive); writes the (u, a, b) updates of consecutive chain members.  The three item lists are CSR over users.
 * qrec_tbpr_sgd_ordered: TBPR.optimization (TBPR.py:40-48) over those triplets strictly in order (a == b occurs and is
 *   applied as two sequential updates of one row), d_loss2[0] = sum(-log s), d_loss2[1] = sum over users of
 *   regU*sum(P*P) + regI*sum(Q*Q) taken after each user's updates (TBPR.py:159); d_sums_in = {sum(P*P), sum(Q*Q)}
 *   of the tables at launch (qrec_sumsq).                                                                            */
int qrec_mt_tbpr_sample_epoch(uint32_t *state625, const int64_t *pos_indptr, const int32_t *pos_items, int32_t n_users,
                              int32_t n_items, const int64_t *joint_indptr, const int32_t *joint_items,
                              const int64_t *weak_indptr, const int32_t *weak_items, const int64_t *strong_indptr,
                              const int32_t *strong_items, int64_t capacity, int32_t *u_out, int32_t *a_out, int32_t *b_out,
                              int64_t *n_out);
int qrec_tbpr_sgd_ordered(void *d_P, void *d_Q, int dtype, int32_t d, int32_t ld, const int32_t *d_u, const int32_t *d_a,
                          const int32_t *d_b, int64_t n, double lr, double regU, double regI, const double *d_sums_in,
                          double *d_loss2, void *stream);

/* ---- SBPR: model/ranking/SBPR.py:31-78 (numpy path) ------------------------------------------------------------------
 * The reference's file raises TypeError at :46 (`self.FPSet[user][kItems]`, a dict indexed with a list) for the first user who has
 * social feedback; both entry points implement the loop with that subscript read as `item_k` and everything else as written
 * (tests/golden/gen_golden.py case_sbpr_filmtrust: the reference's own source with that one token replaced is the fixture).
 * qrec_mt_sbpr_sample_epoch (host): the draws of SBPR.py:37-55,69-72 on the CPython MT19937 stream -> rows (u, i, k or -1, j, Suk),
 *   int32 [n][5].  ps_users: user ids in PositiveSet's key order; fp_*: FPSet as CSR over user ids (items in dict order, counts);
 *   item_key_user[j]: id of the user whose NAME equals item j's name or -1, is_key[user] (in/out): that user is a key of the
 *   defaultdict FPSet by now -- the negative's rejection test `item_j in self.FPSet` (:52) is an item name looked up among user names.
 * qrec_sbpr_sgd_ordered: SBPR.py:41-74 over those rows strictly in order (rows with i < 0: a bare visit of a user without positives).
 *   d_bias: the item biases b [n_items] in the tables' dtype (never updated by the reference); bias_sumsq = b.b;
 *   d_loss2[0] = sum of the -log terms (:57-58, :73), d_loss2[1] = sum over users of regU*sum(P*P) + regI*sum(Q*Q) + b.b (:74);
 *   d_sums_in = {sum(P*P), sum(Q*Q)} of the tables at launch (qrec_sumsq).                                                       */
int qrec_mt_sbpr_sample_epoch(uint32_t *state625, const int32_t *ps_users, int32_t n_ps, const int64_t *pos_indptr, const int32_t *pos_items,
                              int32_t n_users, int32_t n_items, const int64_t *fp_indptr, const int32_t *fp_items, const int32_t *fp_counts,
                              const int32_t *item_key_user, uint8_t *is_key, int64_t capacity, int32_t *rows_out, int64_t *n_out);
int qrec_sbpr_sgd_ordered(void *d_P, void *d_Q, const void *d_bias, int dtype, int32_t d, int32_t ld, const int32_t *d_rows, int64_t n, double lr,
                          double regU, double regI, double bias_sumsq, const double *d_sums_in, double *d_loss2, void *stream);

/* ---- MHCN: model/ranking/MHCN.py:93-216 (self-gating, channel attention, hierarchical self-supervision) ---------------
 * Tables [rows][ld] fp32, ld in {32, 64, 128, 256}, columns >= d zero; d x d weights zero-padded to [ld][ld], biases [ld].
 * qrec_gate_fwd:  Y = X * sigmoid(X W + b), S = sigmoid(.)  -- self_gating / self_supervised_gating (MHCN.py:109-112).
 * qrec_gate_bwd:  Q = (dy_scale dY) * X * S (1 - S);  dX (+)= dy_scale dY * S + Q W^T.  The weight gradients are
 *   dW = X^T Q, db = column sums of Q: qrec_buir_wgrad(X, Q, ...).
 * qrec_channel_attention_fwd (MHCN.py:113-121): v = att_mat att^T (d_v, [ld] scratch kept for the backward pass),
 *   score = softmax_k(e_k . v) ([rows][4], 4th unused), out = sum_k score_k e_k + half / 2 (half nullable).
 * qrec_channel_attention_bwd: de_k (+)= score_k dOut + dw_k v (accumulate flag), dhalf (+)= dOut / 2 (nullable),
 *   g_att += att_mat^T dv, g_att_mat += dv (x) att with dv = sum_rows sum_k dw_k e_k (d_dv_scratch:
 *   qrec_channel_attention_scratch_bytes; the column sums dv, graph, d graph are formed in a fixed order -- per-block partials
 *   added in block order, no float atomics -- so the weight gradients have the same bits on every launch).
 * qrec_hss_loss_grad: hierarchical_self_supervision (MHCN.py:184-206) given edge = H em and the call's five shuffles
 *   (row p1; column k2 then row p2; column k3 then row p3 -- tf.random.shuffle of range(n) / range(d)) with their
 *   inverses: *d_loss += local + global MIM loss (unscaled); d_dem / d_dedge = scale * gradient w.r.t. em / edge
 *   (overwritten).  d_scratch: qrec_hss_scratch_bytes(n).
 * qrec_random_permutations: `count` uniformly random permutations of range(n) ([count][n]) and their inverses from ONE
 *   rocPRIM radix sort (key = permutation number << 40 | 40 Philox bits); qrec_small_permutations: `count` Fisher-Yates permutations of range(n <= 4096) with inverses. */
int qrec_gate_fwd(const float *d_X, const float *d_W, const float *d_bias, int64_t n_rows, int32_t ld, float *d_Y, float *d_S,
                  void *stream);
int qrec_gate_bwd(const float *d_X, const float *d_S, const float *d_dY, const float *d_W, int64_t n_rows, int32_t d, int32_t ld,
                  float dy_scale, float *d_Q, float *d_dX, int32_t accumulate, void *stream);
int qrec_channel_attention_fwd(const float *d_e1, const float *d_e2, const float *d_e3, const float *d_att, const float *d_att_mat,
                               const float *d_half, int64_t n_rows, int32_t ld, float *d_v, float *d_score, float *d_out,
                               void *stream);
int qrec_channel_attention_bwd(const float *d_dOut, const float *d_e1, const float *d_e2, const float *d_e3, const float *d_score,
                               const float *d_v, const float *d_att, const float *d_att_mat, int64_t n_rows, int32_t ld,
                               float *d_de1, float *d_de2, float *d_de3, int32_t accumulate, float *d_dhalf, int32_t half_accumulate,
                               float *d_dv_scratch, float *d_g_att, float *d_g_att_mat, void *stream);
int qrec_channel_attention_scratch_bytes(int64_t *bytes);
int qrec_hss_scratch_bytes(int64_t n_rows, int64_t *bytes);
int qrec_hss_loss_grad(const float *d_em, const float *d_edge, int64_t n_rows, int32_t d, int32_t ld, const int32_t *d_p1,
                       const int32_t *d_p1inv, const int32_t *d_p2, const int32_t *d_p2inv, const int32_t *d_k2,
                       const int32_t *d_k2inv, const int32_t *d_p3, const int32_t *d_p3inv, const int32_t *d_k3,
                       const int32_t *d_k3inv, float scale, float *d_scratch, float *d_dem, float *d_dedge, double *d_loss,
                       void *stream);
int qrec_random_permutations_scratch_bytes(int64_t n, int32_t count, int64_t *bytes);
int qrec_random_permutations(int64_t n, int32_t count, uint64_t seed, uint64_t stream_id, void *d_scratch, int32_t *d_perms,
                             int32_t *d_invs, void *stream);
int qrec_small_permutations(int32_t n, int32_t count, uint64_t seed, uint64_t stream_id, int32_t *d_perms, int32_t *d_invs,
                            void *stream);

/* ---- multi-GPU: collectives and the kernels around them ------------------------------------- *
 * The reference has no distributed code (its only parallelism is one process per CV fold, QRec.py:57-89); the
 * contract is BASELINE.json's north star: one process per GPU, embedding tables row-sharded, RCCL over xGMI.  librccl
 * is bound directly (dlopen'ed on first use from the directory of the HIP runtime this library runs on; QREC_RCCL_LIB
 * overrides) -- no torch tensor or torch kernel is involved.  Every collective only ENQUEUES on `stream`.
 *   qrec_comm_unique_id : rank 0 makes the 128-byte id, the launcher's control plane hands it to every rank;
 *   qrec_comm_init      : collective over all ranks, binds the device selected by qrec_init;
 *   qrec_allreduce      : in-place sum (QREC_F32 / QREC_F64 / QREC_I32);  _pair: two buffers, ONE fused launch (the
 *                         item-table deltas + the epoch's loss terms);
 *   qrec_allgather / qrec_reduce_scatter : `count` elements per rank (graph models: operand rows of a row-partitioned
 *                         propagation and the transposed product of its backward pass);
 *   qrec_alltoall_rows  : segment p of d_send (h_send_rows[p] rows of row_bytes, segments back to back in rank order)
 *                         goes to rank p, segment p of d_recv comes from rank p -- cross-shard row lookups and the
 *                         return of their updates.  Counts are HOST arrays of `world` entries.                       */
#define QREC_COMM_UID_BYTES 128
int qrec_comm_library(char *path_out, int path_len, int *version);
int qrec_comm_unique_id(uint8_t *h_uid);
int qrec_comm_init(int32_t world, int32_t rank, const uint8_t *h_uid, void **comm);
int qrec_comm_destroy(void *comm);
int qrec_comm_info(void *comm, int32_t *world, int32_t *rank);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the ranks that really
 * joined, this process's rank among them and the HIP device it is bound to -- bench.py prints them next to a multi-GPU result */
int qrec_comm_query(void *comm, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *device);
int qrec_allreduce(void *comm, void *d_buf, int64_t count, int dtype, void *stream);
int qrec_allreduce_pair(void *comm, void *d_a, int64_t count_a, int dtype_a, void *d_b, int64_t count_b, int dtype_b,
                        void *stream);
int qrec_allgather(void *comm, const void *d_send, void *d_recv, int64_t count, int dtype, void *stream);
int qrec_reduce_scatter(void *comm, const void *d_send, void *d_recv, int64_t count, int dtype, void *stream);
int qrec_alltoall_rows(void *comm, const void *d_send, const int64_t *h_send_rows, void *d_recv,
                       const int64_t *h_recv_rows, int64_t row_bytes, void *stream);
/* One fused launch of arbitrary point-to-point segments: send k = h_send_bytes[k] bytes at d_send + h_send_off[k] to rank
 * h_send_peer[k]; receive k likewise.  Between two ranks sends and receives are matched in list order (both sides list a pair's
 * segments in the same order).  The row-sharded BPR layout ships the request ids of ALL batches of an epoch with it.       */
int qrec_sendrecv_segments(void *comm, const void *d_send, const int32_t *h_send_peer, const int64_t *h_send_off,
                           const int64_t *h_send_bytes, int32_t n_send, void *d_recv, const int32_t *h_recv_peer,
                           const int64_t *h_recv_off, const int64_t *h_recv_bytes, int32_t n_recv, void *stream);

/* Replicated item table (every rank trains its own users against a full copy of Q): after a step
 *     delta = Q - Q_start;  all-reduce(delta);  Q_start += delta;  Q = Q_start
 * so that every rank's updates are kept and the replicas stay bit-identical.  n = rows*ld floats (multiple of 4).     */
/* The same around the step's ONE collective, fused with the epoch's loss terms (fp32 tables; users sharded, items
 * replicated): qrec_dist_epoch_pre = { delta = Q - Q_start; d_stats[1] = sum P*P of this rank's users }, the caller
 * all-reduces {d_delta, d_stats[0..1]} (qrec_allreduce_pair), qrec_dist_epoch_post = { Q_start += delta; Q = Q_start;
 * d_stats[2] = sum Q*Q; the reference's epoch decision as in qrec_epoch_close }.  Sums are taken in block order:
 * every rank computes the same bits and takes the same decision.                                                    */
int qrec_dist_epoch_pre(const float *d_P, int64_t p_rows, int32_t ld, const float *d_Q, const float *d_Q_start,
                        float *d_delta, int64_t q_rows, double *d_stats, const double *d_state, void *stream);
int qrec_dist_epoch_post(float *d_Q, float *d_Q_start, const float *d_delta, int64_t q_rows, int32_t ld, double *d_stats,
                         double *d_state, double regU, double regI, double max_lr, double tol, double *d_log,
                         int64_t log_capacity, void *stream);
int qrec_table_delta(const float *d_table, const float *d_start, float *d_delta, int64_t n, void *stream);
int qrec_table_apply(float *d_table, float *d_start, const float *d_delta, int64_t n, void *stream);

/* Row-sharded item table: item id = r*world + o is local row r of rank o (interleaved, so that popular items spread
 * over the ranks).  qrec_shard_rows = rows a rank holds.
 * qrec_shard_plan_batch, for n triplets with global item ids d_i / d_j:
 *   d_req_rows : the DISTINCT items the batch touches as local rows at their owner, grouped by owner in rank order and
 *                ascending inside a group (capacity min(2n, n_items));  d_counts[world] = rows per owner;
 *   d_ci, d_cj : the triplets' item ids rewritten as positions in d_req_rows, i.e. rows of the batch's row cache.
 * Everything stays on the device; the caller reads d_counts back when it needs the sizes of the exchange.
 * qrec_shard_plan_epoch: the same plan for ALL n_batches batches of an epoch in one set of launches -- batch b = triplets
 *   [d_bounds[b], d_bounds[b + 1]) (device int64[n_batches + 1]), its request list at d_req_rows + d_req_off[b] (device
 *   int64[n_batches]), its counts at d_counts[b * world]; scratch: qrec_shard_plan_epoch_scratch_bytes.
 * qrec_gather_rows: d_out[k] = d_table[d_rows[k]] (an owner answering a request).
 * qrec_scatter_add_row_deltas: d_table[d_rows[k]] += d_fresh[k] - d_sent[k] with f32 atomics (an owner taking back the
 *   rows it lent: several ranks may return the same row, every rank's updates are kept).                             */
int qrec_shard_rows(int64_t n_items, int32_t world, int32_t rank, int64_t *rows);
int qrec_shard_plan_scratch_bytes(int64_t n_items, int32_t world, int64_t *bytes);
int qrec_shard_plan_batch(const int32_t *d_i, const int32_t *d_j, int64_t n, int64_t n_items, int32_t world,
                          void *d_scratch, int32_t *d_req_rows, int32_t *d_counts, int32_t *d_ci, int32_t *d_cj,
                          void *stream);
int qrec_shard_plan_epoch_scratch_bytes(int64_t n_items, int32_t world, int32_t n_batches, int64_t *bytes);
int qrec_shard_plan_epoch(const int32_t *d_i, const int32_t *d_j, const int64_t *d_bounds, int32_t n_batches, int64_t n,
                          int64_t n_items, int32_t world, void *d_scratch, int32_t *d_req_rows, const int64_t *d_req_off,
                          int32_t *d_counts, int32_t *d_ci, int32_t *d_cj, void *stream);
int qrec_gather_rows(const float *d_table, int32_t ld, const int32_t *d_rows, int64_t n, float *d_out, void *stream);
/* The rows of a training batch out of / into ROW-PARTITIONED tables (graph models; the rows tf.nn.embedding_lookup
 * reads, model/ranking/LightGCN.py:22-24).  Batch row k of 3B is table row u[k] (k < B), n_users + i[k - B] (k < 2B) or
 * n_users + j[k - 2B]; d_block holds the table's rows [lo, hi).
 * qrec_batch_rows_gather:       d_out[k] = the row if lo <= row < hi, else zeros ([3B][ld]): summed over the ranks (one all-reduce of
 *                               3B rows) these are the batch's rows of the whole table -- instead of an all-gather of the table;
 * qrec_batch_rows_scatter_add:  d_block[row(k) - lo] += d_src[k] for every k whose row lies in [lo, hi) (f32 atomics: a batch
 *                               repeats rows; lo = 0, hi = N scatters into a whole table).                                      */
/* ... and for an arbitrary list of table rows d_ids[n] (SimGCL's InfoNCE reads the batch's UNIQUE users / items, SimGCL.py:61-64) */
int qrec_rows_gather_owned(const float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_ids, int64_t n, float *d_out, void *stream);
int qrec_rows_scatter_add_owned(float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_ids, int64_t n, const float *d_src, void *stream);
int qrec_batch_rows_gather(const float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_u, const int32_t *d_i,
                           const int32_t *d_j, int32_t B, int64_t n_users, float *d_out, void *stream);
int qrec_batch_rows_scatter_add(float *d_block, int32_t ld, int64_t lo, int64_t hi, const int32_t *d_u, const int32_t *d_i,
                                const int32_t *d_j, int32_t B, int64_t n_users, const float *d_src, void *stream);
int qrec_scatter_add_row_deltas(float *d_table, int32_t ld, const int32_t *d_rows, int64_t n, const float *d_fresh,
                                const float *d_sent, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QREC_HIP_H */
