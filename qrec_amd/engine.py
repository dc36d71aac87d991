"""Device-side state and epoch drivers of the embedding-training hot path.

``DeviceTables`` keeps the user/item embedding tables resident in HBM for the whole
training run (row-major, row stride ``ld`` padded so one row is a whole number of 64-byte
segments); ``BprSgd`` / ``MfSgd`` run one epoch per call through the C ABI
(include/qrec_hip.h).  Nothing here computes on the host: a missing or failing
libqrec_hip.so raises.
"""
from __future__ import annotations

import numpy as np

from . import capi
from .capi import DeviceBuffer
from .interactions import CSR


def padded_ld(d: int, dtype) -> int:
    """Row stride in elements.  fp32 tables use 32/64/128/256 so that the throughput kernel
    can map a row onto 16/32/64 lanes; fp64 tables (order-exact kernels only) 16/32/64/128 so that the scheduled kernel
    can put a row on the 16 lanes of one DPP row with 16-byte accesses (four triplets per wavefront, bpr_exact.hip);
    wider fp64 rows stay unpadded (one triplet per wavefront).  Pad columns are zero and stay zero under every update."""
    if np.dtype(dtype) == np.float64:
        for ld in (16, 32, 64, 128):
            if d <= ld:
                return ld
        return d
    for ld in (32, 64, 128, 256):
        if d <= ld:
            return ld
    raise ValueError(f"embedding size {d} > 256 is not supported by the fp32 kernels")


class DeviceTables:
    def __init__(self, P: np.ndarray, Q: np.ndarray, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.code = capi.F64 if self.dtype == np.float64 else capi.F32
        self.n_users, self.d = P.shape
        self.n_items = Q.shape[0]
        assert Q.shape[1] == self.d
        self.ld = padded_ld(self.d, self.dtype)
        self.P = DeviceBuffer.from_numpy(self._pad(P))
        self.Q = DeviceBuffer.from_numpy(self._pad(Q))

    def _pad(self, a: np.ndarray) -> np.ndarray:
        out = np.zeros((a.shape[0], self.ld), dtype=self.dtype)
        out[:, :self.d] = a
        return out

    def upload(self, P: np.ndarray, Q: np.ndarray):
        self.P.upload(self._pad(P)); self.Q.upload(self._pad(Q))

    def download(self, dtype=np.float64):
        """(P, Q) as host arrays [rows, d] (pad columns dropped)."""
        return (np.ascontiguousarray(self.P.numpy()[:, :self.d], dtype=dtype),
                np.ascontiguousarray(self.Q.numpy()[:, :self.d], dtype=dtype))

    def sumsq(self, scratch: DeviceBuffer, stream=None):
        """(sum P*P, sum Q*Q) -- model/ranking/BPR.py:40."""
        capi.sumsq(self.P, self.code, self.n_users, self.d, self.ld, scratch, stream)
        p = float(scratch.numpy(stream)[0])
        capi.sumsq(self.Q, self.code, self.n_items, self.d, self.ld, scratch, stream)
        return p, float(scratch.numpy(stream)[0])


def balanced_chunk(n: int, groups: int = 4096, lo: int = 26, hi: int = 40, prefer: int = 32) -> int:
    """Chunk length for the throughput kernels.  Their ``groups`` persistent groups each take ceil(n_chunks/groups)
    or one fewer chunks, and the epoch ends when the busiest group does: with 9.56 chunks per group (Yelp2018 shape at
    chunk 32) 44% of the groups idle through the last chunk.  Pick the length in [lo, hi] whose chunk count divides most
    evenly over the groups (measured: 0.607 ms at chunk 30, 0.583 at 32, 0.577 at 31 and 34)."""
    if n <= 0:
        return prefer
    best, best_cost = prefer, None
    for c in range(lo, hi + 1):
        per = -(-n // c) / groups
        if per <= 1:                       # fewer chunks than groups: nothing to balance
            cost = 0.0
        else:
            cost = -(-per // 1) / per - 1.0      # ceil(per)/per - 1 = idle share of the last round
        key = (round(cost, 4), abs(c - prefer))
        if best_cost is None or key < best_cost:
            best, best_cost = c, key
    return best


# `auto` (QREC_SCHEDULE unset): which throughput schedule an epoch runs under -- ONE-PASS always: "item" when a few items collect most
# interactions (their rows would take the per-triplet atomics of the user-major kernel), else "user" (measured 2.1 vs 1.6 G/s at the
# Zipf-0.6 Yelp2018 shape).
# (The two-pass "deferred negatives" schedule of rounds 3-5 -- one atomic row update per triplet, 0.62 of the roofline on tables that
# live in HBM -- is gone: outside the +-0.002 Recall bar at the last epoch in every setting measured at the size that figure was quoted on,
# profiles/r05_auto_regime_25m.json, r05_fresh_coefficient_25m.json.  Round 6 reaches the same 0.63 with ONE pass: resolve_p_update below.)


def resolve_schedule(n_triplets: int, item_degrees=None, requested: str = "auto"):
    """(schedule, None) for ``BprSgd`` (the second entry was the deferred schedule's sub-epoch count, rounds 3-5); ``item_degrees``: positives per item (None = unknown: treated as skewed)"""
    if requested != "auto":
        return requested, None
    if item_degrees is None:
        return "item", None
    deg = np.asarray(item_degrees)
    return ("item" if deg.size and deg.max() > 20 * max(deg.mean(), 1e-9) else "user"), None


# How P[u] is written by the item-major kernel (round 6).  "atomic": the exact per-sample delta through the L2 atomic units (two atomic
# row updates per triplet: the kernel's bound, 0.43-0.46 of the HBM roofline).  "rmw": sc1 load + sc1 write-through store (QREC_HW_P_RMW,
# include/qrec_hip.h) -- ONE atomic row update per triplet, 0.63-0.64 of the roofline; an update of the same user's row that lands between
# another group's load and store is lost.  How many are: the collision density
#     c = (groups in flight) x sum_u (n_u / n)^2         (expected number of other groups holding the same user while one does)
# Measured, paired Recall@20 against order-exact training (profiles/r06_item_rmw.json; |gap| at the reference's peak / at the last epoch):
#     c = 0.003 .. 0.008  (650 k users, 25 M triplets, d = 128)   rmw 0.0010 / 0.0002, 0.0011 / 0.0002   atomic 0.0008 / 0.0002, 0.0010 / 0.0003
#     c = 0.033           (160 k users, 6 M triplets)              rmw 0.0010 / 0.0010, 0.0018 / 0.0003   atomic 0.0001 / 0.0001, 0.0015 / 0.0005
#     c = 0.17            (Yelp2018 shape, 31.7 k users)           rmw 0.0005 / 0.0009 with a 10 % loss gap   atomic 0.0004 / 0.0000, loss gap 0.2 %
#     c ~ 0.2 .. 2        (lastfm under BPR.conf, 1.9 k users)     rmw mean gap over 16 seeds -0.019 +- 0.001   atomic +0.0000 +- 0.0008
# `auto` therefore takes "rmw" only where c <= P_RMW_MAX_COLLISION = 0.01 -- config #4's shape (10 M users: c = 0.0005), its single-GPU
# slice (0.003), never the Yelp2018 shape, never the reference's own small datasets.
P_RMW_MAX_COLLISION = 0.01
DEFAULT_GROUPS = 4096


def collision_density(user_counts, groups: int = DEFAULT_GROUPS) -> float:
    """groups x sum_u (n_u / n)^2 for the triplets-per-user counts of an epoch"""
    c = np.asarray(user_counts, dtype=np.float64)
    n = float(c.sum())
    return float(groups * np.square(c / n).sum()) if n > 0 else 0.0


def resolve_p_update(user_counts, groups: int = DEFAULT_GROUPS, requested: str = "auto") -> str:
    """"atomic" or "rmw" for the item-major kernel's P[u] updates; ``requested``: "auto" (by collision density), or one of the two"""
    if requested in ("atomic", "rmw"):
        return requested
    if requested != "auto":
        raise ValueError("p_update must be 'auto', 'atomic' or 'rmw'")
    return "rmw" if collision_density(user_counts, groups) <= P_RMW_MAX_COLLISION else "atomic"


MIN_ROUNDS = 8
ITEM_RUN = 16           # item-major stored order: the item-sorted list in runs of this many triplets (BprSgd.__init__)


def grid_for_epoch(n: int, chunk: int, min_rounds: int = MIN_ROUNDS, groups: int = 4096, min_chunk: int = 8):
    """(chunk, groups) of ONE launch over a whole epoch of n triplets such that every persistent group walks at least ``min_rounds``
    chunks one after the other, whatever the epoch's size (round 5).  The kernels' launchers clamp the groups to the number of chunks:
    an epoch with fewer chunks than groups (the reference's lastfm split: 74 k triplets = 2.3 k chunks of 32) is in flight ALL AT ONCE --
    one round of the grid, where the Yelp2018 shape's 1.25 M triplets make ~9.5.  Shorter chunks first (down to ``min_chunk``), then
    fewer groups.  ``groups`` 0 in the result = the launcher's default (nothing to cap).  ``min_rounds`` <= 1: unchanged."""
    if min_rounds <= 1 or n <= 0:
        return chunk, 0
    c = chunk
    while c > min_chunk and -(-n // c) < min_rounds * groups:
        c = max(min_chunk, c // 2)
    n_chunks = -(-n // c)
    if n_chunks >= min_rounds * groups:
        return c, 0
    return c, int(max(1, n_chunks // min_rounds))


def launch_chunk(n: int, chunk: int, groups: int = 4096, lo: int = 4) -> int:
    """Chunk length for ONE launch over n triplets of an epoch that is cut into batches (several ranks: reconciliation / exchange batches).
    A batch of a few ten thousand triplets in chunks of 32 occupies a fraction of the 4,096 persistent groups for one chunk's latency
    (measured, tools/probe_strong_scaling_bound.py: 19.5 k-triplet batches at 8 ranks, 38 us each however small); shorter chunks spread
    the same triplets over the idle groups.  Never longer than the epoch's chunk; at least `lo`."""
    if n >= 2 * groups * chunk:
        return chunk
    return int(max(lo, min(chunk, -(-n // groups))))


def stride_runs(n: int, run: int) -> np.ndarray:
    """positions 0..n-1 cut into runs of ``run`` consecutive ones, the runs in golden-ratio stride order (run r of the result is
    run (r * stride) mod n_runs of the input, stride ~ 0.618 n_runs made coprime): consecutive runs of the result are far apart in
    the input.  A short tail (n mod run positions) goes last, so every run of the result starts at a multiple of ``run``."""
    import math
    n_runs = n // run
    if n_runs == 0:
        return np.arange(n, dtype=np.int64)
    stride = max(int(n_runs * 0.6180339887498949), 1)
    while math.gcd(stride, n_runs) != 1:
        stride += 1
    slots = (np.arange(n_runs, dtype=np.int64) * stride) % n_runs
    at = (slots[:, None] * run + np.arange(run, dtype=np.int64)[None, :]).ravel()
    return np.concatenate([at, np.arange(n_runs * run, n, dtype=np.int64)])


class BprSgd:
    """One BPR epoch per call over a fixed (u, i) triplet list (user-major PositiveSet
    order, model/ranking/BPR.py:31-34); negatives ``j`` come per epoch either from the
    host (exact CPython stream) or from the device Philox sampler.

    Device-side epoch statistics live in one small buffer (sum -log sigma, sum P*P, sum Q*Q, ticket) so
    that an epoch costs a single 24-byte read-back -- or none at all: with ``start_device_driver`` the
    epoch's loss, the convergence test and the bold-driver learning-rate update (BPR.py:40,
    base/iterativeRecommender.py:56-63,88-104) run on the device and epochs are enqueued back to back.  The Philox sampler for
    epoch k+1 runs on a side stream underneath the SGD kernel of epoch k (double-buffered
    negatives); the SGD kernel is bound by the L2 atomic units, the sampler by integer ALU,
    so they overlap almost perfectly."""

    def __init__(self, tables: DeviceTables, u: np.ndarray, i: np.ndarray, pos: CSR | None = None,
                 schedule: str = "user", n_items: int | None = None, batches: int = 1, chunk: int = 32,
                 item_run: int | None = None, p_update: str = "atomic"):
        """``schedule``: "user" keeps the reference's user-major order (required by the order-exact
        kernel); "item" stores the same triplets sorted by positive item for the item-major
        throughput kernel (``self.perm`` maps scheduled position -> reference position).
        ``batches`` > 1 (row-sharded item table, qrec_amd/dist.py): the epoch is trained in that many launches, over
        the triplet ranges ``self.batch_bounds``.  User-major: consecutive ranges.  Item-major: the sorted list is dealt
        to the batches ``chunk`` triplets at a time, round-robin, so that the chunks of one hot item are spread over
        all batches exactly as the kernel's strided visiting order spreads them over a single launch (a batch of
        CONSECUTIVE item-sorted triplets would put all work on a hot row into one launch: measured, 16 % higher loss
        after 14 epochs on the ML-1M shape); batch starts stay multiples of ``chunk``, item runs stay intact."""
        if schedule not in ("user", "item"):
            raise ValueError("schedule must be 'user' or 'item'")
        self.t = tables
        # size of the item catalogue the triplets' ids refer to: the table's rows, except when this process holds only a
        # row shard of the item table (qrec_amd/dist.py) and the ids are global
        self.n_items = int(n_items) if n_items is not None else tables.n_items
        self.n = int(u.size)
        self.schedule = schedule
        self.chunk = int(chunk)             # the chunk the stored order is dealt to the batches in = the chunk to launch epochs with
        # groups a reconciliation batch (replicated layout, K > 1) spreads over: a batch of a few ten thousand triplets is bound by the
        # per-group chain of dependent row reads (~1.9 us per triplet), not by the atomic units -- more groups, shorter chains
        self.batch_groups = 4096
        # ONE launch per epoch: at least MIN_ROUNDS rounds of the grid whatever the epoch's size (grid_for_epoch); `launch_grid()` is what
        # the epoch drivers pass to the kernels
        self._epoch_grid = grid_for_epoch(self.n, self.chunk) if max(1, int(batches)) == 1 else (self.chunk, 0)
        # item-major only: how P[u] is written (resolve_p_update above; "auto" decides by the collision density of THESE triplets under the
        # launcher's default grid).  Explicit "atomic" is the constructor's default: callers opt in to "auto" (the drop-in BPR class and
        # bench.py do), so that a test or a measurement never changes kernels behind the caller's back.
        self.collision = collision_density(np.bincount(np.asarray(u), minlength=1)) if self.n else 0.0
        self.p_update = resolve_p_update(np.bincount(np.asarray(u), minlength=1) if self.n else [], requested=p_update) if schedule == "item" else "atomic"
        self.item_variant = capi.HW_P_RMW if self.p_update == "rmw" else capi.HW_DEFAULT
        self.perm = None
        u = np.ascontiguousarray(u, dtype=np.int32); i = np.ascontiguousarray(i, dtype=np.int32)
        batches = max(1, int(batches))
        per = -(-self.n // batches) if self.n else 0
        self.batch_bounds = [min(b * per, self.n) for b in range(batches + 1)]
        # Item-major stored order (round 4): the item-sorted list is cut into RUNS of `item_run` triplets and the runs are laid out
        # in golden-ratio stride order, so that consecutive runs belong to different items -- a chunk of 32 is two runs of 16.  A
        # positive item's row still rides in registers along its run (one atomic flush per run), but it takes 16 steps in a row
        # instead of 32: measured with NO GPU involved (tools/order_sensitivity.py, profiles/r04_order_sensitivity.json), sequential
        # fp64 training in the 32-run order ends 0.003-0.004 of Recall@20 away from the reference's user-major order on data with
        # structure -- above the +-0.002 bar before any parallel execution -- while the 8-run order stays as close as a random
        # order does; runs of 16 cost nothing measurable either way and are the default (profiles/r04_item_run_timing.txt).  item_run = 0
        # keeps whole item runs (rounds 1-3).  A constructor parameter only: the sampler's negatives depend on the stored order, so an
        # environment variable here would change results behind the caller's back.
        self.item_run = int(item_run if item_run is not None else ITEM_RUN)
        if schedule == "item":
            self.perm = np.argsort(i, kind="stable")
            if self.item_run > 0 and self.n > self.item_run:
                self.perm = self.perm[stride_runs(self.n, self.item_run)]
            if batches > 1 and self.n:
                n_chunks, tail = -(-self.n // chunk), self.n % chunk
                whole = np.arange(n_chunks - 1 if tail else n_chunks)
                dealt = [whole[b::batches] for b in range(batches)]
                order = np.concatenate(dealt + ([np.array([n_chunks - 1])] if tail else []))
                at = (order[:, None] * chunk + np.arange(chunk)[None, :]).ravel()
                self.perm = self.perm[at[at < self.n]]
                sizes = [len(x) * chunk for x in dealt]
                sizes[-1] += tail
                self.batch_bounds = [0] + np.cumsum(sizes).tolist()
            u, i = np.ascontiguousarray(u[self.perm]), np.ascontiguousarray(i[self.perm])
        self.h_u, self.h_i, self.h_j = u, i, None          # host copies: the order-exact mode schedules the epoch on the host
        self._exact = {}
        self.d_u = DeviceBuffer.from_numpy(u)
        self.d_i = DeviceBuffer.from_numpy(i)
        self.d_j = DeviceBuffer(max(self.n, 1), np.int32)
        self.d_j_next = None
        self.d_stats = DeviceBuffer.zeros(capi.STATS_WORDS, np.float64)
        self.d_loss = self.d_stats          # element 0
        self._pos_dev = None
        self._side = None
        self._sampled = None
        self._prefetched_epoch = None
        self._sgd_start = None
        self._own_events = None
        self._consumed = [None, None]        # events: "the SGD kernel that read [d_j, d_j_next] has finished"
        self.d_drv = self.d_log = None
        self._log_capacity = 0
        if pos is not None:
            srt = pos.sorted_rows()
            self._pos_dev = (DeviceBuffer.from_numpy(srt.indptr), DeviceBuffer.from_numpy(srt.indices))

    def launch_grid(self, min_rounds: int | None = None):
        """(chunk, groups) for the whole-epoch launch of the one-pass schedules (groups 0 = the launcher's default grid)"""
        if min_rounds is None:
            return self._epoch_grid
        return grid_for_epoch(self.n, self.chunk, min_rounds) if len(self.batch_bounds) == 2 else (self.chunk, 0)

    # -- negatives ---------------------------------------------------------------------------
    def set_negatives(self, j: np.ndarray, stream=None):
        """``j`` in the reference's (user-major) triplet order; re-ordered to the schedule's."""
        j = np.ascontiguousarray(j, dtype=np.int32)
        if self.perm is not None:
            j = np.ascontiguousarray(j[self.perm])
        self.h_j = j
        self.d_j.upload(j, stream)

    def negatives_reference_order(self) -> np.ndarray:
        """current negatives as a host array in the reference's triplet order"""
        j = self.d_j.numpy()
        if self.perm is None:
            return j
        out = np.empty_like(j); out[self.perm] = j
        return out

    def sample_negatives_device(self, seed: int, epoch: int, stream=None):
        if self._pos_dev is None:
            raise RuntimeError("BprSgd was built without the positives CSR")
        capi.philox_bpr_sample(self._pos_dev[0], self._pos_dev[1], self.d_u, self.n, self.n_items,
                               seed, epoch, self.d_j, stream)

    def prefetch_negatives_device(self, seed: int, epoch: int):
        """Enqueue the sampler for `epoch` on the side stream into the spare buffer."""
        if self._pos_dev is None:
            raise RuntimeError("BprSgd was built without the positives CSR")
        if self._side is None:
            self._side = capi.Stream(); self._sampled = capi.Event()
            self.d_j_next = DeviceBuffer(max(self.n, 1), np.int32)
        if self._consumed[1] is not None:      # the spare buffer may still be read by an enqueued SGD kernel
            capi.stream_wait_event(self._side, self._consumed[1])
        if self._sgd_start is not None:
            capi.stream_wait_event(self._side, self._sgd_start)
        capi.philox_bpr_sample(self._pos_dev[0], self._pos_dev[1], self.d_u, self.n, self.n_items,
                               seed, epoch, self.d_j_next, self._side)
        self._sampled.record(self._side)
        self._prefetched_epoch = epoch

    def take_prefetched_negatives(self, epoch: int, stream=None):
        """Make `stream` wait for the prefetched negatives of `epoch` and switch to them."""
        if self._prefetched_epoch != epoch:
            raise RuntimeError(f"negatives of epoch {epoch} were not prefetched")
        capi.stream_wait_event(stream, self._sampled)
        self.d_j, self.d_j_next = self.d_j_next, self.d_j
        self._consumed.reverse()
        self._prefetched_epoch = None

    def mark_negatives_consumed(self, stream=None):
        """Record, after enqueueing the SGD kernel that reads the current negatives, that their buffer is
        free again once the stream gets here (only needed when epochs are enqueued without host syncs)."""
        if self._consumed[0] is None:
            self._consumed[0] = capi.Event()
        self._consumed[0].record(stream)

    # -- epochs ------------------------------------------------------------------------------------
    def epoch_ordered(self, lr: float, regU: float, regI: float, stream=None, width: int | None = None) -> float:
        """Strictly order-respecting pass (reference semantics).  Returns sum(-log sigmoid).
        The epoch's dependence DAG is list-scheduled on the host (``qrec_bpr_exact_schedule``, ~30 ms per 1.25 M
        triplets) into steps of at most ``width`` independent triplets, executed by one workgroup of ``width`` wavefronts
        (``qrec_bpr_sgd_scheduled``): same values as the one-wavefront walker bit for bit, ~n/6 steps instead of n.
        ``width`` 1 (or env QREC_EXACT_WIDTH=1) selects the walker."""
        import os
        if self.schedule != "user":
            raise RuntimeError("the order-exact kernel needs the reference's user-major order")
        t = self.t
        if width is None:
            width = int(os.environ.get("QREC_EXACT_WIDTH", "0")) or 8     # measured: 8 wavefronts cover the DAG's width (214 k steps vs 209 k at 16) at 2/3 of the step time
        width = min(width, capi.bpr_exact_width(t.code, t.d))
        if width <= 1 or self.n == 0 or self.h_j is None:
            capi.bpr_sgd_ordered(t.P, t.Q, t.code, t.d, t.ld, self.d_u, self.d_i, self.d_j, self.n, lr, regU, regI,
                                 self.d_stats, stream)
            return float(self.d_stats.head(1, stream)[0])
        prep = self.prepare_ordered(self.h_j, width=width, slot=0, stream=stream, reorder=False)
        self.run_prepared(prep, lr, regU, regI, stream)
        return float(self.d_stats.head(1, stream)[0])

    # -- order-exact epochs, pipelined: the host side of epoch k + 1 under the kernel of epoch k ---------------------
    def exact_width(self, width: int | None = None) -> int:
        import os
        if width is None:
            width = int(os.environ.get("QREC_EXACT_WIDTH", "0")) or 8
        return min(width, capi.bpr_exact_width(self.t.code, self.t.d))

    def prepare_ordered(self, j: np.ndarray, width: int | None = None, slot: int = 0, stream=None, reorder: bool = True):
        """Host side of one order-exact epoch: the dependence DAG of (u, i, j) list-scheduled (``qrec_bpr_exact_schedule``)
        and uploaded into schedule buffer ``slot`` (0 / 1) on ``stream``.  Sampling does not look at the embeddings, so
        this can run while the PREVIOUS epoch's kernel is still executing (pass a side stream: a copy on the kernel's
        stream would queue behind it).  ``j``: negatives in the reference's triplet order.  Returns the handle for
        ``run_prepared``; None when the one-wavefront walker has to be used (width <= 1)."""
        if self.schedule != "user":
            raise RuntimeError("the order-exact kernel needs the reference's user-major order")
        t = self.t
        width = self.exact_width(width)
        j = np.ascontiguousarray(j, dtype=np.int32)
        if reorder and self.perm is not None:
            j = np.ascontiguousarray(j[self.perm])
        if width <= 1 or self.n == 0:
            return None
        kind, slots = capi.bpr_exact_kind(t.code, t.ld, width)
        if kind and max(max(t.n_users, t.n_items) * t.ld, self.n) * t.dtype.itemsize >= 0xFFFFFF00:
            kind, slots = 0, 0            # the four-per-wavefront kernel addresses the tables AND its per-triplet log with 32-bit offsets
        entries, off = capi.bpr_exact_schedule(self.h_u, self.h_i, j, t.n_users, self.n_items, width, registers=bool(kind))
        x = self._exact
        if "xlog" not in x:
            x["xlog"] = DeviceBuffer(self.n + capi.EXACT_XLOG_PAD, t.dtype)
            x["scratch"] = DeviceBuffer.zeros(capi.EXACT_SCRATCH_WORDS, np.float64)
        key = f"slot{slot}"
        if key not in x:
            x[key] = (DeviceBuffer((self.n, 8), np.int32), DeviceBuffer(3 * self.n + 2, np.int32))     # step offsets: the register schedule may leave steps empty (<= 3 n steps)
        d_entries, d_off = x[key]
        d_entries.upload(entries, stream)
        d_off.upload_head(off, stream)
        steps = int(off.size - 1)
        # four triplets per wavefront (rows of 16 .. 128 elements): the schedule goes on in the fixed-width layout, expanded on the
        # device behind the upload (same stream)
        d_wide = None
        if kind:
            need = (steps + capi.EXACT_WIDE_PAD) * slots * 8
            wkey = f"wide{slot}"
            if wkey not in x or x[wkey].shape[0] < need:
                x[wkey] = DeviceBuffer(int(need * 1.1) + 64, np.int32)
            d_wide = x[wkey]
            capi.bpr_exact_expand(d_entries, d_off, steps, slots, d_wide, stream)
        # the host arrays stay referenced by the handle: an asynchronous copy may still be reading them when this returns
        return {"entries": d_entries, "off": d_off, "steps": steps, "width": width, "j": j, "host": (entries, off), "slots": slots, "wide": d_wide,
                "kind": kind}

    def run_prepared(self, prep, lr: float, regU: float, regI: float, stream=None):
        """enqueue the scheduled kernel of a prepared epoch (no host synchronisation; the loss lands in the stats buffer)"""
        t, x = self.t, self._exact
        self.h_j = prep["j"]
        self.exact_steps = prep["steps"]
        if prep.get("kind"):
            capi.bpr_sgd_scheduled_wide(t.P, t.Q, t.n_users, t.n_items, t.code, t.d, t.ld, prep["wide"], prep["steps"],
                                        prep["slots"], self.n, lr, regU, regI, x["xlog"], x["scratch"], self.d_stats, stream)
        else:
            capi.bpr_sgd_scheduled(t.P, t.Q, t.code, t.d, t.ld, prep["entries"], prep["off"], prep["steps"], prep["width"], self.n, lr, regU, regI,
                                   x["xlog"], x["scratch"], self.d_stats, stream)

    def epoch_throughput_async(self, lr: float, regU: float, regI: float, chunk: int = 32,
                               variant: int = capi.HW_DEFAULT, stream=None, groups: int = 0, flush_every: int = 16):
        """Hogwild pass (fp32 tables); enqueue only -- read the loss with ``loss()``/``epoch_stats()``.
        User-major schedule: P[u] register-resident, atomics on Q[i], Q[j].  Item-major schedule:
        Q[i] register-resident (flushed + re-read every ``flush_every`` triplets), atomics on P[u],
        Q[j] -- ~20% faster because the atomic units see flatter target rows."""
        if self.t.dtype != np.float32:
            raise TypeError("throughput mode needs fp32 tables")
        capi.memset(self.d_stats.ptr, 0, 8, stream)
        if self.schedule == "item":
            capi.bpr_sgd_hogwild_item_major(self.t.P, self.t.Q, self.t.d, self.t.ld, self.d_u, self.d_i, self.d_j,
                                            self.n, chunk, groups, flush_every, lr, regU, regI, self.d_stats, stream, variant=self.item_variant)
        else:
            capi.bpr_sgd_hogwild(self.t.P, self.t.Q, self.t.d, self.t.ld, self.d_u, self.d_i, self.d_j,
                                 self.n, chunk, groups, lr, regU, regI, self.d_stats, variant, stream)

    # -- device-resident loss / convergence / learning-rate schedule ------------------------------------
    def start_device_driver(self, lr0: float, log_capacity: int = 1024):
        """Put the bold-driver state (include/qrec_hip.h QREC_DRV_*) on the device: lRate = lr0, lastLoss = 0."""
        st = np.zeros(capi.DRV_WORDS, np.float64); st[capi.DRV_LR] = lr0
        self.d_drv = DeviceBuffer.from_numpy(st)
        self.d_log = DeviceBuffer.zeros((max(log_capacity, 1), capi.DRV_LOG_WORDS), np.float64)
        self._log_capacity = log_capacity
        self._own_events = [capi.Event() for _ in range(4)]
        self._grid_events = [capi.Event(), capi.Event()]
        self.d_stats.fill_bytes(0)

    def epoch_device_async(self, regU: float, regI: float, max_lr: float, tol: float = 1e-3, chunk: int = 32,
                           variant: int = capi.HW_DEFAULT, stream=None, groups: int = 0, flush_every: int = 16,
                           events=None, dist=None, after_start=None):
        """One throughput epoch with everything after it (BPR.py:40 loss terms, isConverged,
        updateLearningRate) enqueued on the device: no host synchronisation.  ``events`` = (before, after)
        capi.Event pair recorded around the SGD kernel.  ``dist`` (one process per GPU, qrec_amd/dist.py): a
        ``ReplicatedStep`` or ``ShardedStep`` -- the collectives are enqueued on the same stream, between the kernels.
        ``after_start``: called once the event that releases the next epoch's sampler is recorded (in front of the epoch's first
        SGD grid) -- the place to enqueue that sampler (``prefetch_negatives_device``) when something inside this epoch is to
        wait for it: the sharded layout plans the next epoch in front of this epoch's last batch."""
        if self.d_drv is None:
            raise RuntimeError("call start_device_driver() first")
        t = self.t
        # Two event records per epoch.  `start` (before the SGD grid) releases the next epoch's sampler: it is
        # dispatched just after this epoch's SGD grid -- measured, a sampler that already sits on the CUs when
        # the (one block per CU, persistent) SGD grid arrives skews its placement and costs 30% (0.78 vs 0.59 ms).
        # `end` (after it) frees the negatives buffer for the sampler after next.
        start, end = events if events else (self._own_events[0], self._own_events[1])
        start.record(stream)
        self._sgd_start = start
        if dist is not None and dist.mode == "sharded":
            # the first batch's SGD grid comes behind its fetch (gather, exchange): the sampler must not be released by the epoch's
            # start event -- it would sit on the CUs when the grid arrives (the 30% case above; measured again in round 3, 0.83 vs
            # 0.79 ms/epoch) -- but by one recorded right in front of that grid
            def first_grid(st):
                if after_start is not None:
                    grid = self._grid_events[0]; self._grid_events.reverse()
                    grid.record(st); self._sgd_start = grid
                    after_start()

            dist.exchange.run_epoch(lambda t0, nb, cache, rows, ci, cj, st: self._launch_sgd(
                t.P, cache, self.d_u.ptr + 4 * t0, ci, cj, nb, launch_chunk(nb, chunk), groups, flush_every, regU, regI, variant, st, q_rows=rows), stream,
                next_epoch=(lambda: dist.next_epoch(self)), before_first_sgd=first_grid)
        else:
            if after_start is not None:
                after_start()
            K = len(self.batch_bounds) - 1
            if dist is not None and dist.mode == "replicated" and K > 1:
                # replicated item table reconciled K times per epoch (round 4): between two syncs a rank does not see the other
                # ranks' updates of the rows they share, and with ONE sync per epoch that window is the whole epoch -- measured on
                # data with structure (tools/paired_recall.py, profiles/r04_paired_recall.json) the summed stale deltas overshoot
                # and training diverges at five times BPR.conf's rate.  Batch b = the b-th range of the stored order (item-major:
                # chunks dealt round-robin, so every batch sees every hot item); the last batch's sync is the fused epoch close.
                for b in range(K):
                    t0, nb = self.batch_bounds[b], self.batch_bounds[b + 1] - self.batch_bounds[b]
                    if nb:
                        bg = self.batch_groups
                        self._launch_sgd(t.P, t.Q, self.d_u.ptr + 4 * t0, self.d_i.ptr + 4 * t0, self.d_j.ptr + 4 * t0, nb, launch_chunk(nb, chunk, groups=bg),
                                         groups if groups else (bg if bg != 4096 else 0), flush_every, regU, regI, variant, stream)
                    if b + 1 < K:
                        dist.sync_tables(stream)
            else:
                self._launch_sgd(t.P, t.Q, self.d_u, self.d_i, self.d_j, self.n, chunk, groups, flush_every, regU, regI, variant, stream)
        end.record(stream)
        self._consumed[0] = end
        self._own_events.reverse()
        if dist is None:
            capi.epoch_close(t.P, t.n_users, t.Q, t.n_items, t.code, t.ld, self.d_stats, self.d_drv, regU, regI, max_lr,
                             tol, self.d_log, self._log_capacity, stream)
            return
        if dist.mode == "sharded":      # sum P*P and sum Q*Q are over disjoint row shards: all three terms add over ranks
            capi.epoch_sums(t.P, t.n_users, t.Q, t.n_items, t.code, t.ld, self.d_stats, self.d_drv, stream)
            dist.comm.allreduce(self.d_stats, 3, capi.F64, stream)
        elif dist.sync_p is None:       # users sharded, items replicated: ONE collective carries the deltas and {sum -log sigma, sum P*P}
            sq = dist.sync_q
            capi.dist_epoch_pre(t.P, t.n_users, t.ld, t.Q, sq.start, sq.delta, t.n_items, self.d_stats, self.d_drv, stream)
            dist.comm.allreduce_pair(sq.delta, sq.n, capi.F32, self.d_stats, 2, capi.F64, stream)
            capi.dist_epoch_post(t.Q, sq.start, sq.delta, t.n_items, t.ld, self.d_stats, self.d_drv, regU, regI, max_lr, tol,
                                 self.d_log, self._log_capacity, stream)
            return
        else:                           # both tables replicated (drop-in classes: every rank evaluates from whole tables)
            dist.sync_p.sync(stream)
            dist.sync_q.sync(stream, extra=(self.d_stats, 1, capi.F64))
            capi.epoch_sums(t.P, t.n_users, t.Q, t.n_items, t.code, t.ld, self.d_stats, self.d_drv, stream)
        capi.epoch_decide(self.d_stats, self.d_drv, regU, regI, max_lr, tol, self.d_log, self._log_capacity, stream)

    def _launch_sgd(self, P, Q, d_u, d_i, d_j, n, chunk, groups, flush_every, regU, regI, variant, stream, q_rows=None):
        """the throughput kernel of the schedule on (P, Q) -- Q is the item table, or a shard's row cache with the
        triplets' item ids rewritten to its rows; learning rate and stop flags come from the device-side driver"""
        t = self.t
        if self.schedule == "item":
            capi.bpr_sgd_hogwild_item_major(P, Q, t.d, t.ld, d_u, d_i, d_j, n, chunk, groups, flush_every, 0.0, regU, regI,
                                            self.d_stats, stream, self.d_drv, p_rows=t.n_users, q_rows=q_rows, variant=self.item_variant)
        else:
            capi.bpr_sgd_hogwild(P, Q, t.d, t.ld, d_u, d_i, d_j, n, chunk, groups, 0.0, regU, regI, self.d_stats, variant,
                                 stream, self.d_drv, p_rows=t.n_users, q_rows=q_rows)

    def driver_state(self, stream=None) -> dict:
        s = self.d_drv.numpy(stream)
        return {"lr": float(s[capi.DRV_LR]), "last_loss": float(s[capi.DRV_LAST_LOSS]), "epochs": int(s[capi.DRV_EPOCHS]),
                "converged": bool(s[capi.DRV_CONVERGED]), "failed": bool(s[capi.DRV_FAILED])}

    def driver_log(self, stream=None) -> np.ndarray:
        """rows {loss, lr used, sum(-log sigma), lastLoss - loss, sum P*P, sum Q*Q, -, -} of the epochs closed so far"""
        n = min(self.driver_state(stream)["epochs"], self._log_capacity)
        return self.d_log.numpy(stream)[:n].copy()

    def enqueue_epoch_stats(self, stream=None):
        """sum P*P and sum Q*Q into the stats buffer, next to the epoch's sum(-log sigma) -- BPR.py:40,53"""
        t = self.t
        capi.sumsq(t.P, t.code, t.n_users, t.d, t.ld, self.d_stats.ptr + 8, stream)
        capi.sumsq(t.Q, t.code, t.n_items, t.d, t.ld, self.d_stats.ptr + 16, stream)

    def read_epoch_stats(self, stream=None):
        s = self.d_stats.head(3, stream)
        return float(s[0]), float(s[1]), float(s[2])

    def epoch_stats(self, stream=None):
        """(sum -log sigma, sum P*P, sum Q*Q) after the enqueued epoch.  One read-back; this is the epoch's
        only host synchronisation."""
        self.enqueue_epoch_stats(stream)
        return self.read_epoch_stats(stream)

    def loss(self, stream=None) -> float:
        return float(self.d_stats.head(1, stream)[0])


class MfSgd:
    """Rating-prediction MF family on the device, order-exact: BasicMF (model/rating/BasicMF.py:9-26),
    PMF (model/rating/PMF.py:9-28), SVD (model/rating/SVD.py:13-35), EE (model/rating/EE.py:15-34)."""

    def __init__(self, tables: DeviceTables, n: int, variant: int = capi.MF_BASIC, Bu=None, Bi=None):
        self.t = tables
        self.n = n
        self.variant = variant
        self.d_u = DeviceBuffer(max(n, 1), np.int32)
        self.d_i = DeviceBuffer(max(n, 1), np.int32)
        self.d_r = DeviceBuffer(max(n, 1), np.float64)
        self.d_stats = DeviceBuffer.zeros(5, np.float64)      # err^2, sum P^2, sum Q^2, sum Bu^2, sum Bi^2
        self.d_Bu = self.d_Bi = None
        if variant in (capi.MF_SVD, capi.MF_EE):
            self.d_Bu = DeviceBuffer.from_numpy(np.ascontiguousarray(Bu, dtype=tables.dtype))
            self.d_Bi = DeviceBuffer.from_numpy(np.ascontiguousarray(Bi, dtype=tables.dtype))

    def epoch(self, u, i, r, lr: float, regU: float = 0.0, regI: float = 0.0, regB: float = 0.0,
              global_mean: float = 0.0, stream=None) -> float:
        """one pass over the rows in the given order; returns sum(error^2)"""
        self.d_u.upload(np.ascontiguousarray(u, dtype=np.int32), stream)
        self.d_i.upload(np.ascontiguousarray(i, dtype=np.int32), stream)
        self.d_r.upload(np.ascontiguousarray(r, dtype=np.float64), stream)
        capi.mf_sgd_ordered(self.t.P, self.t.Q, self.t.code, self.t.d, self.t.ld, self.d_u, self.d_i,
                            self.d_r, self.n, lr, self.d_stats, stream, self.variant, regU, regI, self.d_Bu,
                            self.d_Bi, regB, global_mean)
        return float(self.d_stats.head(1, stream)[0])

    def sumsq_terms(self, stream=None):
        """(sum P^2, sum Q^2, sum Bu^2, sum Bi^2) for the epoch-end regularisers (PMF.py:25, SVD.py:32-33)"""
        t = self.t
        capi.sumsq(t.P, t.code, t.n_users, t.d, t.ld, self.d_stats.ptr + 8, stream)
        capi.sumsq(t.Q, t.code, t.n_items, t.d, t.ld, self.d_stats.ptr + 16, stream)
        if self.d_Bu is not None:
            capi.sumsq(self.d_Bu, t.code, t.n_users, 1, 1, self.d_stats.ptr + 24, stream)
            capi.sumsq(self.d_Bi, t.code, t.n_items, 1, 1, self.d_stats.ptr + 32, stream)
        s = self.d_stats.numpy(stream)
        return float(s[1]), float(s[2]), float(s[3]), float(s[4])

    def biases(self):
        return self.d_Bu.numpy().astype(np.float64), self.d_Bi.numpy().astype(np.float64)


class SvdppSgd:
    """SVD++ on the device, order-exact (model/rating/SVDPlusPlus.py:25-62): P, Q, the implicit-feedback table Y and
    the biases stay resident; ``rated`` = every user's train items in ``data.userRated`` order."""

    def __init__(self, tables: DeviceTables, Y, Bu, Bi, rated: CSR, n: int):
        self.t, self.n = tables, n
        dt = tables.dtype
        Yp = np.zeros((Y.shape[0], tables.ld), dtype=dt); Yp[:, :tables.d] = Y
        self.d_Y = DeviceBuffer.from_numpy(Yp)
        self.d_Bu = DeviceBuffer.from_numpy(np.ascontiguousarray(Bu, dtype=dt))
        self.d_Bi = DeviceBuffer.from_numpy(np.ascontiguousarray(Bi, dtype=dt))
        self.d_indptr = DeviceBuffer.from_numpy(rated.indptr.astype(np.int64))
        self.d_items = DeviceBuffer.from_numpy(rated.indices.astype(np.int32) if rated.indices.size else np.zeros(1, np.int32))
        self.d_u = DeviceBuffer(max(n, 1), np.int32); self.d_i = DeviceBuffer(max(n, 1), np.int32)
        self.d_r = DeviceBuffer(max(n, 1), np.float64)
        self.d_stats = DeviceBuffer.zeros(6, np.float64)     # err^2, sum P^2, Q^2, Y^2, Bu^2, Bi^2

    def epoch(self, u, i, r, lr, regU, regI, regB, regY, global_mean, stream=None) -> float:
        self.d_u.upload(np.ascontiguousarray(u, dtype=np.int32), stream)
        self.d_i.upload(np.ascontiguousarray(i, dtype=np.int32), stream)
        self.d_r.upload(np.ascontiguousarray(r, dtype=np.float64), stream)
        t = self.t
        capi.svdpp_sgd_ordered(t.P, t.Q, self.d_Y, self.d_Bu, self.d_Bi, t.code, t.d, t.ld, self.d_indptr, self.d_items,
                               self.d_u, self.d_i, self.d_r, self.n, lr, regU, regI, regB, regY, global_mean, self.d_stats, stream)
        return float(self.d_stats.head(1, stream)[0])

    def sumsq_terms(self, stream=None):
        """(sum P^2, sum Q^2, sum Y^2, sum Bu^2, sum Bi^2) -- SVDPlusPlus.py:64-65"""
        t = self.t
        for k, (buf, rows, d, ld) in enumerate(((t.P, t.n_users, t.d, t.ld), (t.Q, t.n_items, t.d, t.ld), (self.d_Y, t.n_items, t.d, t.ld),
                                                (self.d_Bu, t.n_users, 1, 1), (self.d_Bi, t.n_items, 1, 1))):
            capi.sumsq(buf, t.code, rows, d, ld, self.d_stats.ptr + 8 * (k + 1), stream)
        s = self.d_stats.numpy(stream)
        return tuple(float(x) for x in s[1:6])

    def download(self):
        return (self.d_Y.numpy()[:, :self.t.d].astype(np.float64), self.d_Bu.numpy().astype(np.float64),
                self.d_Bi.numpy().astype(np.float64))
