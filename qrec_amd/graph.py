"""Graph side of the LightGCN-family hot path: the joint normalized adjacency
(base/graphRecommender.py:10-29), its segmented-CSR launch plan, and the per-batch training
step of model/ranking/LightGCN.py:11-41 driven entirely through the C ABI."""
from __future__ import annotations

import contextlib
import os
import threading

import numpy as np

from . import capi
from .capi import DeviceBuffer
from .engine import padded_ld

# ---- reduction order of the batch-gradient scatters ------------------------------------------------------------------------
# Throughput mode adds a batch's row gradients into the table gradient with float atomics (right to fp32 rounding; the order,
# and with it the last bits, change from launch to launch).  Parity mode -- the exact mode of the drop-in classes, the golden-run
# tests -- adds every row's lookups in batch order, one lookup at a time, as the reference's CPU scatter does
# (csrc/ordered.hip): same inputs, same bits, every run.  A trainer takes the setting of the thread that constructs it.
_REDUCTIONS = threading.local()


@contextlib.contextmanager
def ordered_reductions(on: bool = True):
    """``with ordered_reductions(): tr = SEPTTrainer(...)`` -- trainers built inside run their gradient scatters in parity mode"""
    before = getattr(_REDUCTIONS, "ordered", False)
    _REDUCTIONS.ordered = bool(on)
    try:
        yield
    finally:
        _REDUCTIONS.ordered = before


def _ordered_ws():
    return capi.OrderedScatter() if getattr(_REDUCTIONS, "ordered", False) else None


def joint_norm_adjacency(n_users: int, n_items: int, uid: np.ndarray, iid: np.ndarray):
    """CSR (indptr int64, indices int32, values float32) of  D^-1/2 (R + R^T) D^-1/2  over
    the N = n_users + n_items nodes, with the reference's arithmetic: entries of R are
    float32 ones and duplicated (u,i) rows add up, degrees are float32 row sums,
    d = float32(rowsum ** -0.5) with inf -> 0, and each value is fl32(fl32(d_r * a) * d_c)
    (base/graphRecommender.py:15-28).  Columns ascending inside each row."""
    import scipy.sparse as sp
    n = n_users + n_items
    uid = np.asarray(uid, dtype=np.int32); iid = np.asarray(iid, dtype=np.int32) + np.int32(n_users)
    rows = np.concatenate([uid, iid]); cols = np.concatenate([iid, uid])
    # structure by scipy's counting-sort COO -> CSR (duplicates summed: float32 sums of ones, exact), columns sorted per row;
    # 0.09 s at the Yelp2018 shape where np.unique over 2.5 M (row, col) keys + a weighted bincount took 0.28-0.5 s -- this runs
    # twice per epoch in SGL / BUIR and once in SEPT.  The values are formed below with the reference's float32 arithmetic.
    M = sp.coo_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(n, n)).tocsr()
    M.sort_indices()
    a = M.data
    indptr = M.indptr.astype(np.int64); c = M.indices.astype(np.int32)
    cnt = np.diff(indptr)
    rowsum = np.zeros(n, np.float32)
    nz = np.flatnonzero(cnt)
    if nz.size:
        rowsum[nz] = np.add.reduceat(a, indptr[:-1][nz])        # integer-valued float32 entries: the sum is exact in any order
    with np.errstate(divide="ignore"):
        d_inv = np.power(rowsum, -0.5)                          # float32 ** python float -> float32
    d_inv[np.isinf(d_inv)] = 0.0
    r = np.repeat(np.arange(n, dtype=np.int64), cnt)
    vals = (d_inv[r] * a).astype(np.float32) * d_inv[c]
    return indptr, c, vals.astype(np.float32)


def sample_subgraph_edges(state625: np.ndarray, uid: np.ndarray, iid: np.ndarray, n_users: int, n_items: int,
                          aug_type: int, drop_rate: float):
    """The kept edges of one augmented sub-graph, drawn like model/ranking/SGL.py:118-135 from the CPython
    generator (``state625`` is advanced in place):
      aug_type 0 (node dropout): random.sample of int(U*rate) users, then of int(I*rate) items; an edge
                 survives iff both end points do;
      aug_type 1 / 2 (edge dropout / random walk): random.sample of int(E*(1-rate)) edge positions of the
                 CURRENT trainingData order.
    Returns (uid_kept, iid_kept); duplicates stay in (they add up in the adjacency, as in the reference)."""
    if drop_rate <= 0:
        return uid, iid
    if aug_type == 0:
        drop_u = capi.mt_sample_range(state625, n_users, int(n_users * drop_rate))
        drop_i = capi.mt_sample_range(state625, n_items, int(n_items * drop_rate))
        keep_u = np.ones(n_users, bool); keep_u[drop_u] = False
        keep_i = np.ones(n_items, bool); keep_i[drop_i] = False
        keep = keep_u[uid] & keep_i[iid]
        return uid[keep], iid[keep]
    if aug_type in (1, 2):
        keep_idx = capi.mt_sample_range(state625, uid.size, int(uid.size * (1 - drop_rate)))
        return uid[keep_idx], iid[keep_idx]
    raise ValueError("aug_type must be 0 (node dropout), 1 (edge dropout) or 2 (random walk)")


class SubgraphSampler:
    """Per-epoch graph augmentation on the device (SGL.py:113-155, BUIR.py:41-65; the throughput mode of SGL and BUIR, csrc/augment.hip):
    a sub-graph is a VALUE array over the full graph's CSR structure, so the full graph's SpMM plan serves every sub-graph
    (``SpmmPlan.with_values``) and nothing is rebuilt.  Host work, once: where each training row's two entries sit in the CSR, the row of
    every CSR entry, numpy's own float32 power(k, -0.5) table.  Per draw: one device permutation (the exact-size random subset of
    random.sample, from the throughput mode's Philox stream) + two small kernels.

    ``draw(aug_type, drop_rate, seed, stream_id)`` -> DeviceBuffer float32[nnz]; the stream is a function of (seed, stream_id) alone:
      aug 1 / 2 (edge dropout / random walk): kept rows = the first int(E (1 - rate)) entries of permutation(E; seed, stream_id)
      aug 0 (node dropout): dropped users = the first int(U rate) entries of permutation(U; seed, stream_id), dropped items = the first
                            int(I rate) of permutation(I; seed, stream_id + 1)."""

    def __init__(self, n_users: int, n_items: int, uid: np.ndarray, iid: np.ndarray, adj):
        indptr, indices, _ = adj
        self.nu, self.ni, self.n = int(n_users), int(n_items), int(n_users + n_items)
        self.n_edges, self.nnz = int(uid.size), int(indices.size)
        if self.nnz >= 2 ** 31 or self.n_edges >= 2 ** 31:
            raise ValueError("SubgraphSampler: 32-bit CSR positions")
        uid = np.asarray(uid, np.int64); iid = np.asarray(iid, np.int64)
        row_of = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(indptr))
        csr_keys = row_of * self.n + indices                                    # ascending: rows ascending, columns ascending inside a row
        pos_ui = np.searchsorted(csr_keys, uid * self.n + (iid + self.nu))
        pos_iu = np.searchsorted(csr_keys, (iid + self.nu) * self.n + uid)
        assert np.array_equal(csr_keys[pos_ui], uid * self.n + iid + self.nu) and np.array_equal(csr_keys[pos_iu], (iid + self.nu) * self.n + uid)
        self.max_deg = int(np.bincount(np.concatenate([uid, iid + self.nu]), minlength=self.n).max()) if self.n_edges else 0
        with np.errstate(divide="ignore"):
            dinv = np.power(np.arange(self.max_deg + 1, dtype=np.float32), -0.5)  # float32 ** python float -> float32: the reference's d_inv
        dinv[np.isinf(dinv)] = 0.0
        up = DeviceBuffer.from_numpy
        self.d_u, self.d_i = up(uid.astype(np.int32)), up(iid.astype(np.int32))
        self.d_pos_ui, self.d_pos_iu = up(pos_ui.astype(np.int32)), up(pos_iu.astype(np.int32))
        self.d_row_of, self.d_indices, self.d_dinv = up(row_of.astype(np.int32)), up(np.asarray(indices, np.int32)), up(dinv.astype(np.float32))
        self.d_cnt, self.d_deg = DeviceBuffer(max(self.nnz, 1), np.int32), DeviceBuffer(max(self.n, 1), np.int32)
        self.d_flags = DeviceBuffer(max(self.n, 1), np.uint8)
        m = max(self.n_edges, self.nu, self.ni, 1)
        self.scratch = DeviceBuffer(capi.random_permutations_scratch_bytes(m, 1), np.uint8)
        self.d_perm, self.d_perm2 = DeviceBuffer(m, np.int32), DeviceBuffer(max(self.ni, 1), np.int32)

    def sizes(self, aug_type: int, drop_rate: float):
        """(kept rows) for edge dropout / random walk, (dropped users, dropped items) for node dropout: SGL.py:118-130's int() truncations"""
        if aug_type == 0:
            return int(self.nu * drop_rate), int(self.ni * drop_rate)
        return (int(self.n_edges * (1 - drop_rate)),)

    def draw(self, aug_type: int, drop_rate: float, seed: int, stream_id: int, out: DeviceBuffer | None = None, stream=None) -> DeviceBuffer:
        out = out if out is not None else DeviceBuffer(max(self.nnz, 1), np.float32)
        common = (self.d_u, self.d_i, self.d_pos_ui, self.d_pos_iu, self.n_edges, self.nu, self.ni, self.d_row_of, self.d_indices, self.nnz,
                  self.d_dinv, self.max_deg, self.d_cnt, self.d_deg, out)
        if drop_rate <= 0:
            capi.subgraph_values(*common, stream=stream)
        elif aug_type == 0:
            ku, ki = self.sizes(0, drop_rate)
            capi.random_permutations(self.nu, 1, seed, stream_id, self.scratch, self.d_perm, None, stream)
            capi.random_permutations(self.ni, 1, seed, stream_id + 1, self.scratch, self.d_perm2, None, stream)
            capi.subgraph_values(*common, d_drop_users=self.d_perm, n_drop_users=ku, d_drop_items=self.d_perm2, n_drop_items=ki,
                                 d_flags=self.d_flags, stream=stream)
        elif aug_type in (1, 2):
            (k,) = self.sizes(aug_type, drop_rate)
            capi.random_permutations(self.n_edges, 1, seed, stream_id, self.scratch, self.d_perm, None, stream)
            capi.subgraph_values(*common, d_keep_rows=self.d_perm, n_keep=k, stream=stream)
        else:
            raise ValueError("aug_type must be 0 (node dropout), 1 (edge dropout) or 2 (random walk)")
        return out


def spectral_row_key(indptr: np.ndarray, indices: np.ndarray, values: np.ndarray, split_row: int, n_iter: int = 12,
                     seed: int = 0) -> np.ndarray:
    """A 1-D embedding of the nodes of a bipartite, symmetrically normalised adjacency in which the nodes of one community
    sit together: power iteration on A^2 (users -> users, items -> items) with the two trivial eigenvectors
    (sqrt(degree) on either side) projected out, i.e. a vector from the span of the leading non-trivial eigenvectors.
    With planted communities that span is block-constant per community, so sorting by the key lines the communities
    up; on a structureless graph it is noise and costs nothing but ~0.15 s of host time."""
    import scipy.sparse as sp
    n = indptr.size - 1
    nnz_row = np.diff(indptr)
    A = sp.csr_matrix((values.astype(np.float64), indices, indptr), shape=(n, n))
    matvec = A.dot                               # 25 products: 0.12 s at the Yelp2018 shape (gather + np.add.reduceat: 0.47 s)

    def unit_sqrt_deg(lo, hi):
        t = np.zeros(n); t[lo:hi] = np.sqrt(nnz_row[lo:hi])
        return t / max(np.linalg.norm(t), 1e-30)

    tu, ti = unit_sqrt_deg(0, split_row), unit_sqrt_deg(split_row, n)
    x = np.zeros(n)
    x[:split_row] = np.random.Generator(np.random.PCG64(seed)).standard_normal(split_row)
    for _ in range(n_iter):                      # user side only: A^2 is block diagonal, the two sides would drift apart
        x -= tu * (tu @ x)
        x = matvec(matvec(x))
        x[split_row:] = 0.0
        x /= max(np.linalg.norm(x), 1e-300)
    x -= tu * (tu @ x)
    y = matvec(x)                                # the matching item-side singular vector
    y[:split_row] = 0.0
    y -= ti * (ti @ y)
    x = x + y / max(np.linalg.norm(y), 1e-300)
    return x / np.sqrt(np.maximum(nnz_row, 1))      # D^-1/2 v: the random-walk form, comparable across degrees


class SpmmPlan:
    """Device-resident CSR plus its segment decomposition (include/qrec_hip.h, qrec_spmm_csr)."""

    def __init__(self, indptr: np.ndarray, indices: np.ndarray, values: np.ndarray, ld: int, seg_len: int = 128,
                 split_row: int | None = None, chunks: int | None = None, row_chunk: np.ndarray | None = None):
        """``split_row`` (bipartite joint adjacency: the number of users): rows below it only gather operand rows
        at or above it and vice versa, so the two kinds of rows are dealt to different XCDs -- workgroups go round-robin
        over the 8 XCDs, each with its own 4 MiB L2, and the kernel is bound by L2 misses: an XCD that only
        runs user rows caches only the item half of the operand.
        ``chunks`` (1, 2 or 4, bipartite plans only; default 4): the rows of each side are put in
        the order of a spectral key (``spectral_row_key``: rows of one community end up next to each other) and cut
        into that many runs of equal non-zeros, each run on XCDs of its own -- on a graph with community structure an
        XCD then gathers mostly its own communities' operand rows (measured: -20 % on a planted-community graph,
        nothing lost on a structureless one).  Only the order of the segment list changes: same results.
        ``row_chunk``: the row -> run map of another plan over the same nodes (``plan.row_chunk``): per-epoch sub-graphs
        (SGL, BUIR) reuse the full graph's map instead of paying the 0.24 s of power iterations for every new plan."""
        n_rows = indptr.size - 1
        nnz_row = np.diff(indptr)
        n_seg_row = np.maximum(1, -(-nnz_row // seg_len)).astype(np.int64)     # ceil, >=1 (empty rows write zeros)
        seg_row = np.repeat(np.arange(n_rows, dtype=np.int32), n_seg_row)
        first_seg = np.concatenate([[0], np.cumsum(n_seg_row)[:-1]])
        k_in_row = np.arange(seg_row.size, dtype=np.int64) - np.repeat(first_seg, n_seg_row)
        seg_beg = indptr[seg_row] + k_in_row * seg_len
        seg_len_arr = np.minimum(seg_len, indptr[seg_row.astype(np.int64) + 1] - seg_beg).astype(np.int32)
        is_long = np.repeat(n_seg_row > 1, n_seg_row)
        seg_slot = np.full(seg_row.size, -1, dtype=np.int32)
        seg_slot[is_long] = np.arange(int(is_long.sum()), dtype=np.int32)
        long_rows = np.nonzero(n_seg_row > 1)[0].astype(np.int32)
        long_count = n_seg_row[long_rows].astype(np.int32)
        long_first = np.concatenate([[0], np.cumsum(long_count)[:-1]]).astype(np.int32) if long_rows.size else np.zeros(0, np.int32)
        # longest segments first: the heavy work starts early, the tail is made of short rows.  (Round 6 measured the alternative the locality
        # argument suggests -- inside an XCD's class the segments in spectral-key order, rows of one community next to each other in time:
        # 68.8 -> 87.7 us on the structureless graph, 57.6 -> 76.5 us on the planted-community one, profiles/r06_spmm_order.jsonl: neighbouring
        # groups then ask for the same operand lines at the same moment, and the launch is ~2 rounds of the grid deep, so "in time" means little.)
        order = np.argsort(-seg_len_arr, kind="stable")
        bipartite = split_row is not None and 0 < split_row < n_rows
        if row_chunk is not None and (not bipartite or row_chunk.size != n_rows):
            raise ValueError("SpmmPlan: row_chunk needs split_row and one entry per row")
        if chunks is None:
            chunks = (int(row_chunk.max()) + 1 if row_chunk is not None and row_chunk.size else
                      4 if bipartite and indices.size > 0 else 1)
        if chunks not in (1, 2, 4) or (chunks > 1 and not bipartite):
            raise ValueError(f"SpmmPlan: chunks must be 1, 2 or 4 and needs split_row (got {chunks})")
        self.row_chunk = None
        if bipartite:
            cls = (seg_row[order] >= split_row).astype(np.int64)
            if chunks > 1:
                self.row_chunk = row_chunk if row_chunk is not None else self._row_chunks(indptr, indices, values, split_row, chunks)
                cls = cls * chunks + self.row_chunk[seg_row[order]]
            order = order[self._deal_by_xcd(cls, 2 * chunks, 4 * (64 // (ld // 4)))]
        self.n_rows, self.nnz, self.ld, self.chunks = n_rows, int(indices.size), ld, chunks
        self.n_segs, self.n_long = int(seg_row.size), int(long_rows.size)
        up = DeviceBuffer.from_numpy
        self.seg_row, self.seg_beg = up(seg_row[order]), up(seg_beg[order].astype(np.int64))
        self.seg_len, self.seg_slot = up(seg_len_arr[order]), up(seg_slot[order])
        self.long_row = up(long_rows) if self.n_long else None
        self.long_first = up(long_first) if self.n_long else None
        self.long_count = up(long_count) if self.n_long else None
        self.partial = DeviceBuffer((max(int(is_long.sum()), 1), ld), np.float32)
        self.indices, self.values = up(indices.astype(np.int32)), up(values.astype(np.float32))

    def with_values(self, d_values: DeviceBuffer) -> "SpmmPlan":
        """this plan's structure (segments, XCD dealing, index array -- shared, not copied) over another value array: a sub-graph of
        the same graph drawn on the device (SubgraphSampler), dropped entries 0"""
        import copy
        if int(np.prod(d_values.shape)) < max(self.nnz, 1):
            raise ValueError("with_values: value array shorter than the plan's non-zeros")
        p = copy.copy(self)
        p.values = d_values
        return p

    @staticmethod
    def _row_chunks(indptr, indices, values, split_row: int, chunks: int) -> np.ndarray:
        """chunk id (0..chunks-1) of every row: each side's rows in spectral-key order, cut at equal non-zero counts"""
        key = spectral_row_key(indptr, indices, values, split_row)
        nnz_row = np.diff(indptr)
        out = np.zeros(nnz_row.size, np.int64)
        for lo, hi in ((0, split_row), (split_row, nnz_row.size)):
            o = lo + np.argsort(key[lo:hi], kind="stable")
            c = np.cumsum(nnz_row[o])
            out[o] = np.minimum(chunks - 1, (c - 1) * chunks // max(int(c[-1]), 1))
        return out

    @staticmethod
    def _deal_by_xcd(cls: np.ndarray, n_cls: int, groups_per_block: int) -> np.ndarray:
        """Entry p of the list is processed by workgroup (p // groups_per_block) of the persistent grid (a multiple
        of 8 workgroups), which runs on XCD (p // groups_per_block) % 8.  The 8 XCDs are divided evenly among the
        ``n_cls`` classes (2, 4 or 8: operand side x row chunk); the positions of a class' XCDs receive that
        class' entries in their given (longest-first) order; whatever does not fit (unequal class sizes) fills the
        remaining positions.  Returns the permutation ``deal``: new list = old list[deal]."""
        n = cls.size
        pos_cls = ((np.arange(n) // groups_per_block) % 8) * n_cls // 8
        deal = np.empty(n, dtype=np.int64)
        free = np.ones(n, bool)
        rest = []
        for c in range(n_cls):
            ents = np.nonzero(cls == c)[0]
            slots = np.nonzero(pos_cls == c)[0]
            k = min(ents.size, slots.size)
            deal[slots[:k]] = ents[:k]; free[slots[:k]] = False
            rest.append(ents[k:])
        deal[np.nonzero(free)[0]] = np.concatenate(rest)
        return deal

    def bytes_algorithmic(self, d: int) -> int:
        """nnz*(4+4) + 4*(N+1) + 2*N*d*4 (every dense row read once, written once)."""
        return self.nnz * 8 + 4 * (self.n_rows + 1) + 2 * self.n_rows * d * 4


class LightGCNTrainer:
    """E = [U;V] resident in HBM; one ``train_step`` = forward propagation (L SpMMs with the
    layer sum fused in), batch BPR loss + gradient scatter, backward propagation (L SpMMs,
    A symmetric), dense TF-1.14 Adam."""

    def __init__(self, U0: np.ndarray, V0: np.ndarray, adj, n_layers: int, lr: float, reg: float,
                 loss_eps: float = 1e-7):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.L, self.lr, self.reg, self.loss_eps = n_layers, lr, reg, loss_eps
        self.plan = SpmmPlan(adj[0], adj[1], adj[2], self.ld, split_row=self.nu)
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        self.E = DeviceBuffer.from_numpy(E0)
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.m, self.v, self.S, self.dE, self.A, self.B = z(), z(), z(), z(), z(), z()
        self.d_loss = DeviceBuffer.zeros(1, np.float64)
        self.row_mask = DeviceBuffer.zeros((self.n + 31) // 32, np.uint32)   # non-zero rows of the batch gradient
        self.batch_rows = None  # capi.RowSubset: the same rows as a list
        self.t = 0
        f = np.float32
        self.b1, self.b2, self.adam_eps = f(0.9), f(0.999), f(1e-8)
        self.b1p, self.b2p = self.b1, self.b2          # fp32 beta powers, as TF keeps them
        self.dp = None                                  # dist.BatchParallel: this rank trains a share of every step's rows

    # ---- pieces -----------------------------------------------------------------------------
    def forward_sum(self, stream=None, last_rows=None):
        """S = E0 + E1 + ... + EL  (divide by L+1 at the point of use).  ``last_rows`` (row bitmap): the last
        layer is only computed at those rows -- a training step reads S at the batch's rows and nowhere else
        (embedding_lookup, LightGCN.py:22-24), and E_L feeds nothing further."""
        x = self.E
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            capi.spmm_csr(self.plan, x, y, self.ld, d_accum=self.S, stream=stream, d_accum_init=self.E if k == 0 else None,   # S = E + A E, no copy
                          d_y_row_mask=last_rows if k == self.L - 1 else None)
            x = y
        if self.L == 0:
            self.S.copy_from(self.E, stream)

    def backward_from_dE(self, stream=None):
        """H_L with H_0 = dE, H_{k+1} = dE + A H_k; the gradient of E is H_L / (L+1)."""
        x = self.dE
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            # the first operand is the batch gradient itself: <= 3B non-zero rows, the rest is skipped
            capi.spmm_csr(self.plan, x, y, self.ld, d_addend=self.dE, addend_scale=1.0, stream=stream,
                          d_x_row_mask=self.row_mask if k == 0 else None, d_addend_row_mask=self.row_mask)
            x = y
        return x

    def adam_alpha(self) -> float:
        """lr * sqrt(1 - beta2_power) / (1 - beta1_power), evaluated in fp32 (TF ApplyAdam)."""
        f = np.float32
        return float(f(f(self.lr) * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p)))

    def train_step_async(self, d_u, d_i, d_j, B: int, stream=None):
        """u/i/j: device pointers (int32[B]); enqueue only, read the loss with ``loss()``."""
        # one launch: the bitmap of the rows {u, nu+i, nu+j} of the batch, cleared first, and the loss accumulator cleared
        bound = min(3 * B, self.n)
        if self.batch_rows is None or self.batch_rows.capacity < bound:
            self.batch_rows = capi.RowSubset(bound)
        subset = capi.mark_compact_batch_rows(d_u, d_i, d_j, B, self.nu, self.n, self.row_mask, self.batch_rows, bound, stream,
                                              d_zero8=self.d_loss, n_zero8=1)
        self.forward_sum(stream, last_rows=self.row_mask)
        # dE is written at the batch's rows (loss scatter) and read at the batch's rows only (operand mask of the first
        # backward product, addend mask of all of them): clearing those ~6 k rows replaces a 17.8 MB fill
        if self.L == 0:      # no backward product: Adam reads dE itself, every row of it -- rows of earlier batches must be gone
            self.dE.fill_bytes(0, stream)
        else:
            capi.zero_rows(self.dE, self.ld, subset, stream)
        if B:
            capi.bpr_batch_loss_grad(self.S, float(self.L + 1), self.nu, self.n, self.ld, d_u, d_i, d_j, B,
                                     self.loss_eps, self.reg, self.dE, self.d_loss, stream, d_row_mask=self.row_mask, ordered=self.ows)
        g = self.backward_from_dE(stream)
        if self.dp is not None:        # the step's gradient and loss = sums over the ranks' shares of its rows
            self.dp.all_reduce(g); self.dp.all_reduce(self.d_loss)
        capi.adam_step(self.E, self.m, self.v, g, self.n * self.ld, 1.0 / (self.L + 1), self.adam_alpha(),
                       float(self.b1), float(self.b2), float(self.adam_eps), stream)
        self.last_grad = (g, 1.0 / (self.L + 1), 0.0)
        self.t += 1
        self.b1p = np.float32(self.b1p * self.b1); self.b2p = np.float32(self.b2p * self.b2)

    def loss(self, stream=None) -> float:
        return float(self.d_loss.numpy(stream)[0])

    def gradients(self):
        """the last step's gradient of (U, V) as the reference's minimize() applies it (before Adam)"""
        g = _applied_gradient(self.last_grad)[:, :self.d]
        return g[:self.nu], g[self.nu:]

    def final_embeddings(self):
        """(U, V) = split(mean(E0..EL)) as float32 host arrays (LightGCN.py:41)."""
        self.forward_sum()
        Ebar = (self.S.numpy()[:, :self.d] / np.float32(self.L + 1)).astype(np.float32)
        return np.ascontiguousarray(Ebar[:self.nu]), np.ascontiguousarray(Ebar[self.nu:])

    def ego_embeddings(self):
        E = self.E.numpy()
        return E[:self.nu, :self.d].copy(), E[self.nu:, :self.d].copy()


def _setup_row_exchange(tr, adj, blk_ptr, sel):
    """How a row-partitioned trainer gets the operand rows of a propagation product (env QREC_GRAPH_EXCHANGE):
    ``allgather`` (default): every rank receives every block, N x ld floats per product and rank;
    ``referenced`` (round 3; SURVEY s8e "all-to-all of only referenced remote rows"): a rank receives the distinct remote rows its
    block of the adjacency refers to -- worked out once per graph (dist.RowPartition.reference), one row gather + one row
    all-to-all per product into a compact operand whose column numbering the block's CSR is rewritten to.  Same products,
    same bits; fewer bytes where the graph has locality (DESIGN.md s7 has both synthetic graphs)."""
    tr.exchange = os.environ.get("QREC_GRAPH_EXCHANGE", "allgather")
    if tr.exchange not in ("allgather", "referenced"):
        raise ValueError("QREC_GRAPH_EXCHANGE must be allgather or referenced")
    if tr.exchange == "referenced":
        cols = tr.rp.reference(adj[0], adj[1])
        tr.plan_ref = SpmmPlan(blk_ptr, cols, adj[2][sel], tr.ld)
        tr.X_ref = DeviceBuffer.zeros((tr.rp.ref_rows, tr.ld), np.float32)


def _row_operand(tr, x, stream):
    """(plan, operand) for  y = A_hat[lo:hi] X  with this rank's block ``x`` of X: the exchange of the operand rows goes out here"""
    if tr.exchange == "referenced":
        tr.rp.gather_referenced(x, tr.X_ref, stream)
        return tr.plan_ref, tr.X_ref
    tr.rp.gather_operand(x, tr.X_full, stream)
    return tr.plan, tr.X_full


class RowPartitionedLightGCNTrainer:
    """LightGCN with the propagation ROW-PARTITIONED over the ranks: rank r owns the contiguous
    rows [lo, hi) of the joint adjacency, of E = [U; V], of the Adam slots and of every propagated layer.  One step at
    the reference's batch size:

        forward   per layer: all-gather of the operand blocks -> the rank's rows of  A_hat X  (same SpMM kernel, same
                  segment order, so each row is the bits the single-GPU product gives); layer sum kept per block;
        loss      all-gather of the layer sum S, then every rank evaluates the whole batch (B rows, replicated: it is
                  ~1 % of a step) and keeps its rows of dE;
        backward  H_0 = dE, H_{k+1} = dE + A_hat H_k  -- A_hat is symmetric, so the backward product is again
                  all-gather + the rank's rows (no transposed product, no reduce-scatter);
        Adam      on the rank's rows.

    Unlike ``dist.BatchParallel`` the step is the SAME step as on one GPU (same batch, same update), the SpMM work per
    rank is 1/G, and no rank holds more than its rows of E, m, v (the gathered operand is a transient).  The price was
    2L + 1 all-gathers of N x ld floats per step: at the Yelp2018 shape (17.8 MB each, L = 3: 125 MB per step and rank
    against a 0.38 ms single-GPU step) the links lose to one GPU's cache hierarchy; it is the layout for graphs whose
    operand no longer fits one GPU's caches or memory.  DESIGN.md s7 has the numbers.

    Round 4 -- the batch-row economies of the single-GPU step (``batch_rows=True``, the default):
      * the loss reads the layer sum at the batch's <= 3B rows only (embedding_lookup, LightGCN.py:22-24): every rank contributes its
        rows of those 3B (zeros elsewhere) and ONE all-reduce of 3B x ld floats (1.5 MB at B = 2048) replaces the all-gather of S;
      * every rank evaluates the whole batch, so every rank HOLDS the whole batch gradient (<= 3B non-zero rows): the first backward
        product takes it as its operand directly -- no exchange at all -- with the row mask that skips the zero rows;
      * the last forward layer is computed at the rank's batch rows only (its other rows feed nothing).
    2L - 1 operand exchanges + 1.5 MB instead of 2L + 1: L = 3 at the Yelp2018 shape 125 -> 92 MB per step and rank (all-gather),
    72 -> 54 MB (referenced rows)."""

    def __init__(self, comm, U0: np.ndarray, V0: np.ndarray, adj, n_layers: int, lr: float, reg: float,
                 loss_eps: float = 1e-7, batch_rows: bool = True):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        from .dist import RowPartition
        self.comm = comm
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.L, self.lr, self.reg, self.loss_eps = n_layers, lr, reg, loss_eps
        self.batch_rows = bool(batch_rows)
        self._batch = {}
        rp = self.rp = RowPartition(comm, self.n, self.ld)
        lo, hi, pad = rp.lo, rp.hi, rp.rows_pad
        indptr, indices, values = adj
        blk_ptr = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64)
        blk_ptr = np.concatenate([blk_ptr, np.full(pad - (hi - lo), blk_ptr[-1] if hi > lo else 0, np.int64)])   # pad rows: empty
        sel = slice(int(indptr[lo]), int(indptr[hi]))
        self.plan = SpmmPlan(blk_ptr, indices[sel], values[sel], self.ld)     # rows = this rank's block, columns global
        _setup_row_exchange(self, adj, blk_ptr, sel)
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        blk = np.zeros((pad, self.ld), np.float32); blk[:hi - lo] = E0[lo:hi]
        self.E = DeviceBuffer.from_numpy(blk)
        zb = lambda: DeviceBuffer.zeros((pad, self.ld), np.float32)
        self.m, self.v, self.S, self.A, self.B = zb(), zb(), zb(), zb(), zb()
        full = lambda: DeviceBuffer.zeros((rp.world * pad, self.ld), np.float32)
        self.X_full, self.S_full, self.dE_full = full(), full(), full()
        self.d_loss = DeviceBuffer.zeros(1, np.float64)
        f = np.float32
        self.b1, self.b2, self.adam_eps = f(0.9), f(0.999), f(1e-8)
        self.b1p, self.b2p = self.b1, self.b2

    def _propagate(self, x, addend, accum, stream, whole_first=None, first_x_mask=None, last_y_mask=None):
        """L layers from the block ``x``: y = A_hat[lo:hi] gather(x) (+ addend), accumulated into ``accum`` if given.
        ``whole_first``: the first operand is already whole on every rank (the batch gradient): no exchange for that product;
        ``first_x_mask``: row bitmap of its non-zero rows; ``last_y_mask``: the last product only at these rows of the block."""
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            if k == 0 and whole_first is not None:
                plan, X = self.plan, whole_first
            else:
                plan, X = _row_operand(self, x, stream)
            capi.spmm_csr(plan, X, y, self.ld, d_addend=addend, addend_scale=1.0 if addend is not None else 0.0,
                          d_accum=accum, stream=stream, d_x_row_mask=first_x_mask if (k == 0 and whole_first is not None) else None,
                          d_y_row_mask=last_y_mask if k == self.L - 1 else None)
            x = y
        return x

    def forward_sum(self, stream=None, last_y_mask=None):
        self.S.copy_from(self.E, stream)
        self._propagate(self.E, None, self.S, stream, last_y_mask=last_y_mask)

    def _batch_buffers(self, B: int):
        """per batch size: the 3B-row tables of the compact loss and the index arrays that address them as (u, i, j) = (k, k, B + k)"""
        b = self._batch.get(B)
        if b is None:
            ar = np.arange(B, dtype=np.int32)
            b = self._batch[B] = dict(S=DeviceBuffer.zeros((3 * max(B, 1), self.ld), np.float32), dE=DeviceBuffer.zeros((3 * max(B, 1), self.ld), np.float32),
                                      u=DeviceBuffer.from_numpy(ar if B else np.zeros(1, np.int32)), j=DeviceBuffer.from_numpy(ar + B if B else np.zeros(1, np.int32)))
        return b

    def train_step_async(self, d_u, d_i, d_j, B: int, stream=None):
        rp, ld = self.rp, self.ld
        if not self.batch_rows or B == 0 or self.L == 0:
            return self._train_step_full(d_u, d_i, d_j, B, stream)
        if getattr(self, "row_mask", None) is None:
            self.row_mask = DeviceBuffer.zeros((rp.world * rp.rows_pad + 31) // 32, np.uint32)
        self.row_mask.fill_bytes(0, stream)
        capi.mark_batch_rows(d_u, d_i, d_j, B, self.nu, self.row_mask, stream)         # bitmap of {u, nu + i, nu + j} over the whole table
        blk_mask = self.row_mask.ptr + (rp.lo // 32) * 4                                # the block starts on a word boundary (rows_pad % 32 == 0)
        self.forward_sum(stream, last_y_mask=blk_mask)
        b = self._batch_buffers(B)
        # the batch's rows of S: every rank its own (zeros elsewhere), summed over the ranks
        capi.batch_rows_gather(self.S, ld, rp.lo, rp.hi, d_u, d_i, d_j, B, self.nu, b["S"], stream)
        self.comm.allreduce(b["S"], 3 * B * ld, capi.F32, stream)
        b["dE"].fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
        # the loss on the compact table: batch row k is the user, B + k the positive, 2B + k the negative item of triplet k (n_users = B)
        capi.bpr_batch_loss_grad(b["S"], float(self.L + 1), B, 3 * B, ld, b["u"], b["u"], b["j"], B, self.loss_eps, self.reg, b["dE"], self.d_loss, stream, ordered=self.ows)
        # every rank holds the whole batch gradient: scattered into a whole-height operand (repeated rows add up), which the first
        # backward product reads directly; this rank's rows of it are the addend of all of them
        self.dE_full.fill_bytes(0, stream)
        capi.batch_rows_scatter_add(self.dE_full, ld, 0, self.n, d_u, d_i, d_j, B, self.nu, b["dE"], stream)
        dE_blk = self.dE_full.ptr + rp.lo * ld * 4
        g = self._propagate(dE_blk, dE_blk, None, stream, whole_first=self.dE_full, first_x_mask=self.row_mask)
        self._adam(g, stream)

    def _adam(self, g, stream):
        rp, ld = self.rp, self.ld
        f = np.float32
        alpha = float(f(f(self.lr) * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p)))
        capi.adam_step(self.E, self.m, self.v, g, (rp.hi - rp.lo) * ld, 1.0 / (self.L + 1), alpha, float(self.b1), float(self.b2),
                       float(self.adam_eps), stream)
        self.b1p = f(self.b1p * self.b1); self.b2p = f(self.b2p * self.b2)

    def _train_step_full(self, d_u, d_i, d_j, B: int, stream=None):
        """the step of rounds 2-3: every layer in full, the layer sum all-gathered, the batch gradient exchanged like any operand"""
        rp, ld = self.rp, self.ld
        self.forward_sum(stream)
        rp.gather_operand(self.S, self.S_full, stream)
        self.dE_full.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
        if B:
            capi.bpr_batch_loss_grad(self.S_full, float(self.L + 1), self.nu, self.n, ld, d_u, d_i, d_j, B, self.loss_eps,
                                     self.reg, self.dE_full, self.d_loss, stream, ordered=self.ows)
        dE_blk = self.dE_full.ptr + rp.lo * ld * 4                       # this rank's rows of the batch gradient, in place
        g = self._propagate(dE_blk, dE_blk, None, stream)
        self._adam(g, stream)

    def loss(self, stream=None) -> float:
        return float(self.d_loss.numpy(stream)[0])

    def final_embeddings(self):
        """(U, V) = split(mean(E0..EL)) (LightGCN.py:41), whole, on every rank"""
        self.forward_sum()
        self.rp.gather_operand(self.S, self.S_full)
        Ebar = (self.S_full.numpy()[:self.n, :self.d] / np.float32(self.L + 1)).astype(np.float32)
        return np.ascontiguousarray(Ebar[:self.nu]), np.ascontiguousarray(Ebar[self.nu:])

    def block(self, buf) -> np.ndarray:
        """this rank's rows of a block buffer (pad rows and pad columns dropped)"""
        return buf.numpy()[:self.rp.hi - self.rp.lo, :self.d].copy()


class BprTfTrainer:
    """BPR.trainModel_tf (model/ranking/BPR.py:77-96): batch loss
    -sum log(sigmoid(y) + 1e-6) + reg*(l2_loss(U) + l2_loss(V)) over the FULL tables, Adam.
    Same kernels as LightGCN with zero propagation layers; the full-table L2 gradient reg*theta is
    folded into the Adam kernel and its loss term comes from qrec_sumsq."""

    def __init__(self, U0, V0, lr: float, reg: float, loss_eps: float = 1e-6):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.lr, self.reg, self.loss_eps = lr, reg, loss_eps
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        self.E = DeviceBuffer.from_numpy(E0)
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.m, self.v, self.dE = z(), z(), z()
        self.d_loss = DeviceBuffer.zeros(2, np.float64)      # [batch term, sum theta^2]
        f = np.float32
        self.b1, self.b2, self.adam_eps = f(0.9), f(0.999), f(1e-8)
        self.b1p, self.b2p = self.b1, self.b2

    def train_step_async(self, d_u, d_i, d_j, B: int, stream=None):
        f = np.float32
        self.dE.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
        capi.bpr_batch_loss_grad(self.E, 1.0, self.nu, self.n, self.ld, d_u, d_i, d_j, B, self.loss_eps, 0.0,
                                 self.dE, self.d_loss, stream, ordered=self.ows)
        capi.sumsq(self.E, capi.F32, self.n, self.d, self.ld, self.d_loss.ptr + 8, stream)   # loss is of the pre-update tables
        alpha = float(f(f(self.lr) * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p)))
        capi.adam_step(self.E, self.m, self.v, self.dE, self.n * self.ld, 1.0, alpha, float(self.b1), float(self.b2),
                       float(self.adam_eps), stream, grad_l2=self.reg)
        self.last_grad = (self.dE, 1.0, self.reg)
        self.b1p = f(self.b1p * self.b1); self.b2p = f(self.b2p * self.b2)

    def loss(self, stream=None) -> float:
        batch, ss = self.d_loss.numpy(stream)
        return float(batch + self.reg * 0.5 * ss)

    def tables(self):
        E = self.E.numpy()
        return E[:self.nu, :self.d].copy(), E[self.nu:, :self.d].copy()

    def gradients(self, before):
        """``before`` = [U; V] when the step started ([n, d]): the full-table L2 term of BPR.py:83 is reg * theta"""
        g = _applied_gradient((self.last_grad[0], 1.0, 0.0))[:, :self.d] + np.float32(self.reg) * np.asarray(before, np.float32)
        return g[:self.nu], g[self.nu:]


def _applied_gradient(last, before=None):
    """Host copy of what a step handed to Adam, BEFORE the update: ``last`` = (gradient buffer, scale, l2 coefficient) as the
    trainers leave it after ``train_step_async`` (the buffers survive until the next step).  The reference's gradient is
    scale * buffer + l2 * theta, theta the variable when the step started (``before``, padded like the buffer; needed only
    when l2 != 0: the Adam kernel forms that term from theta itself).  Parity tests compare this with the gradients the
    reference's ``minimize`` applied (tests/golden/tf_*.npz: grad<k>_<var>), before Adam's normalisation amplifies noise."""
    buf, scale, l2 = last
    g = buf.numpy().astype(np.float32) * np.float32(scale)
    if l2:
        if before is None:
            raise ValueError("this gradient has a full-table L2 part: pass the variable's value at the start of the step")
        g = g + np.float32(l2) * np.asarray(before, np.float32)
    return g


def unique_first_appearance(idx: np.ndarray) -> np.ndarray:
    """tf.unique(x)[0]: distinct values in order of first occurrence (SimGCL.py:61-64)."""
    _, first = np.unique(idx, return_index=True)
    return idx[np.sort(first)]


class SimGCLTrainer:
    """model/ranking/SimGCL.py:15-111 on the device: a clean LightGCN encoder (no ego layer in
    the mean) plus two noise-perturbed ones, batch BPR loss on the clean view, InfoNCE between
    the perturbed views on the batch's unique users / items, one shared backward pass (the
    three encoders have the same linear backward operator, so their output gradients are
    summed first: L SpMMs instead of 3L), dense TF-1.14 Adam."""

    def __init__(self, U0, V0, adj, n_layers: int, lr: float, reg: float, cl_rate: float, eps: float,
                 tau: float = 0.2, loss_eps: float = 1e-7, seed: int = 0, max_unique: int = 4096):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.L, self.lr, self.reg, self.cl_rate, self.eps, self.tau = n_layers, lr, reg, cl_rate, eps, tau
        self.loss_eps, self.seed = loss_eps, seed
        self.plan = SpmmPlan(adj[0], adj[1], adj[2], self.ld, split_row=self.nu)
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        self.E = DeviceBuffer.from_numpy(E0)
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.m, self.v = z(), z()
        self.Sm, self.S1, self.S2, self.dOut, self.A, self.B = z(), z(), z(), z(), z(), z()
        self.V = [[z(), z()], [z(), z()]]                      # ping-pong layer buffers of the two perturbed views
        self.d_loss = DeviceBuffer.zeros(2, np.float64)         # [rec, cl (unscaled)]
        self.row_mask = DeviceBuffer.zeros((self.n + 31) // 32, np.uint32)   # non-zero rows of dOut (batch rows)
        self.max_unique = max_unique
        self.ws = DeviceBuffer(capi.info_nce_workspace_bytes(max_unique, self.ld), np.uint8)
        # the user-side and item-side InfoNCE terms are independent chains of small kernels: the item side runs on a
        # second HIP stream (own workspace), forked and joined with events -- no host synchronisation
        self.ws_items = DeviceBuffer(capi.info_nce_workspace_bytes(max_unique, self.ld), np.uint8)
        self.side_stream, self.ev_fork, self.ev_join = capi.Stream(), capi.Event(), capi.Event()
        self.batch_rows = None  # capi.RowSubset of the step's rows
        f = np.float32
        self.b1, self.b2, self.adam_eps = f(0.9), f(0.999), f(1e-8)
        self.b1p, self.b2p = self.b1, self.b2
        self.step_no = 0
        self.dp = None          # dist.BatchParallel

    def _encode(self, S, view: int, noises=None, stream=None, last_rows=None):
        """S = sum_k emb_k over the L propagated layers (view 0 = clean; 1, 2 = perturbed).  ``last_rows``: row
        bitmap of the batch -- the last layer is computed there only (the lookups of SimGCL.py:53-55,66-69 read
        nothing else; rows outside the bitmap keep stale values that nobody reads)."""
        S.fill_bytes(0, stream)
        x = self.E
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            ymask = last_rows if k == self.L - 1 else None
            if view == 0:
                capi.spmm_csr(self.plan, x, y, self.ld, d_accum=S, stream=stream, d_y_row_mask=ymask)
            else:
                capi.spmm_csr(self.plan, x, y, self.ld, stream=stream, d_y_row_mask=ymask)
                noise = None if noises is None else noises[(view - 1) * self.L + k]
                capi.perturb_rows(y, self.n, self.d, self.ld, self.eps, noise, self.seed,
                                  (self.step_no * 2 + (view - 1)) * 64 + k, d_accum=S, stream=stream)
            x = y

    def _encode_three(self, noises, stream, last_rows, last_subset=None):
        """The clean and the two perturbed encoders of one training step (SimGCL.py:23-36).  All three start from the
        same E, so the first product A E is formed once and perturbed twice in one pass, which also STARTS the three layer
        sums (no zero-fill); from the second layer on each view has its own operand: 3L-2 SpMMs instead of 3L (L=2: 1 full
        + 3 row-masked instead of 3 + 3).  The last layer is computed -- and perturbed -- at the batch's rows only."""
        x = [self.E, self.E, self.E]
        sums = [self.Sm, self.S1, self.S2]
        nz = lambda v, k: None if noises is None else noises[(v - 1) * self.L + k]
        sid = lambda v, k: (self.step_no * 2 + (v - 1)) * 64 + k
        for k in range(self.L):
            last = k == self.L - 1
            ymask, rows = (last_rows, last_subset) if last else (None, None)
            y0 = self.A if k % 2 == 0 else self.B
            if k == 0:
                capi.spmm_csr(self.plan, self.E, y0, self.ld, stream=stream, d_y_row_mask=ymask)
                y1, y2 = self.V[0][0], self.V[1][0]
                capi.perturb_two_views(y0, y1, y2, self.n, self.d, self.ld, self.eps, nz(1, 0), nz(2, 0), self.seed, sid(1, 0), sid(2, 0),
                                       self.S1, self.S2, self.Sm, stream, rows=rows)
                x = [y0, y1, y2]
                continue
            capi.spmm_csr(self.plan, x[0], y0, self.ld, d_accum=self.Sm, stream=stream, d_y_row_mask=ymask)
            for v in (1, 2):
                yv = self.V[v - 1][k % 2]
                capi.spmm_csr(self.plan, x[v], yv, self.ld, stream=stream, d_y_row_mask=ymask)
                capi.perturb_rows(yv, self.n, self.d, self.ld, self.eps, nz(v, k), self.seed, sid(v, k), d_accum=sums[v],
                                  stream=stream, rows=rows)
                x[v] = yv
            x[0] = y0

    def adam_alpha(self) -> float:
        f = np.float32
        return float(f(f(self.lr) * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p)))

    def train_step_async(self, d_u, d_i, d_j, B: int, d_uniq_users, n_uu: int, d_uniq_items, n_ui: int,
                         noises=None, stream=None, share=None):
        """u/i/j int32[B]; d_uniq_users: distinct user rows; d_uniq_items: distinct item rows
        ALREADY offset by n_users; noises: optional list of 2L device buffers [N][ld] (tests).
        ``share`` = (offset, count), data-parallel runs: the BPR term covers this rank's rows of the step only; the
        InfoNCE terms (which couple all the step's unique rows) are formed by every rank on the whole step with
        weight cl_rate / world, so the all-reduced gradient carries them exactly once."""
        if max(n_uu, n_ui) > self.max_unique:
            raise ValueError("more unique rows in the batch than the InfoNCE workspace holds")
        L = float(self.L)
        lo, cnt = (0, B) if share is None else share
        cl_rate = self.cl_rate if self.dp is None else self.cl_rate / self.dp.world
        bound = min(3 * B, self.n)
        if self.batch_rows is None or self.batch_rows.capacity < bound:
            self.batch_rows = capi.RowSubset(bound)
        subset = capi.mark_compact_batch_rows(d_u, d_i, d_j, B, self.nu, self.n, self.row_mask, self.batch_rows, bound, stream,
                                              d_zero8=self.d_loss, n_zero8=2)          # ... and both loss accumulators cleared
        self._encode_three(noises, stream, self.row_mask, subset)
        capi.zero_rows(self.dOut, self.ld, subset, stream)      # written and read at the batch's rows only (masks below)
        # the InfoNCE rows (unique batch users / positive items) are a subset of the rows marked above
        if share is not None:
            d_u, d_i, d_j = (capi.device_ptr(p) + 4 * lo for p in (d_u, d_i, d_j))
        if cnt:
            capi.bpr_batch_loss_grad(self.Sm, L, self.nu, self.n, self.ld, d_u, d_i, d_j, cnt,
                                     self.loss_eps, self.reg, self.dOut, self.d_loss, stream, d_row_mask=self.row_mask, ordered=self.ows)
        cl = self.d_loss.ptr + 8
        side = self.side_stream.handle
        self.ev_fork.record(stream); self.side_stream.wait_event(self.ev_fork)
        capi.info_nce_loss_grad(self.S1, self.S2, L, d_uniq_items, n_ui, self.ld, self.tau, cl_rate, self.ws_items,
                                self.dOut, cl, side)          # item rows: disjoint from the user rows written below
        capi.info_nce_loss_grad(self.S1, self.S2, L, d_uniq_users, n_uu, self.ld, self.tau, cl_rate, self.ws,
                                self.dOut, cl, stream)
        self.ev_join.record(side); capi.stream_wait_event(stream, self.ev_join)
        # dE0 = (1/L) sum_{k=1..L} A^k dOut :  W_0 = dOut, W_k = dOut + A W_{k-1}, G = A W_{L-1}
        x = self.dOut
        for k in range(self.L - 1):
            y = self.A if k % 2 == 0 else self.B
            capi.spmm_csr(self.plan, x, y, self.ld, d_addend=self.dOut, addend_scale=1.0, stream=stream,
                          d_x_row_mask=self.row_mask if k == 0 else None, d_addend_row_mask=self.row_mask)   # operand = sparse batch gradient
            x = y
        g = self.B if x is self.A else self.A
        capi.spmm_csr(self.plan, x, g, self.ld, stream=stream, d_x_row_mask=self.row_mask if self.L == 1 else None)
        if self.dp is not None:
            self.dp.all_reduce(g)
            self.dp.all_reduce(self.d_loss.head_view(1))      # rec term: shares add up; the cl term is whole on every rank
        capi.adam_step(self.E, self.m, self.v, g, self.n * self.ld, 1.0 / L, self.adam_alpha(), float(self.b1),
                       float(self.b2), float(self.adam_eps), stream)
        self.last_grad = (g, 1.0 / L, 0.0)
        self.b1p = np.float32(self.b1p * self.b1); self.b2p = np.float32(self.b2p * self.b2)
        self.step_no += 1

    def losses(self, stream=None):
        """(total, rec_loss, cl_loss) like the reference prints them (SimGCL.py:104-107)."""
        rec, cl = self.d_loss.numpy(stream)
        cl *= self.cl_rate
        return float(rec + cl), float(rec), float(cl)

    def gradients(self):
        g = _applied_gradient(self.last_grad)[:, :self.d]
        return g[:self.nu], g[self.nu:]

    def main_embeddings(self):
        self._encode(self.Sm, 0)
        m = (self.Sm.numpy()[:, :self.d] / np.float32(self.L)).astype(np.float32)
        return np.ascontiguousarray(m[:self.nu]), np.ascontiguousarray(m[self.nu:])

    def ego_embeddings(self):
        E = self.E.numpy()
        return E[:self.nu, :self.d].copy(), E[self.nu:, :self.d].copy()


class RowPartitionedSimGCLTrainer:
    """SimGCL with every node table ROW-PARTITIONED over the ranks (BASELINE config #5).  Rank r owns rows
    [lo, hi) of E and its Adam slots, of the adjacency, and of the three encoders' layer outputs and layer sums.  One step
    at the reference's batch size, the same step as ``SimGCLTrainer``:

        forward   per layer and encoder: all-gather of the operand blocks -> the rank's rows of A_hat X -> (views 1, 2) the
                  perturbation of those rows, noise keyed by the TABLE row; the first product A_hat E is shared by the three
                  encoders as on one GPU;
        loss      the three layer sums are all-gathered; every rank evaluates the batch's BPR term and both InfoNCE terms on
                  the whole batch (a few thousand rows) and keeps its rows of the gradient;
        backward  the three encoders share one linear backward operator: W_0 = dOut, W_k = dOut + A_hat W_{k-1}, each product
                  all-gather + the rank's rows (A_hat symmetric);
        Adam      on the rank's rows.

    1 + 3 (L - 1) + 3 + L all-gathers of N x ld floats per step (L = 2: nine of 17.8 MB at the Yelp2018 shape)."""

    def __init__(self, comm, U0, V0, adj, n_layers: int, lr: float, reg: float, cl_rate: float, eps: float,
                 tau: float = 0.2, loss_eps: float = 1e-7, seed: int = 0, max_unique: int = 4096, batch_rows: bool = True):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        from .dist import RowPartition
        self.comm = comm
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.L, self.lr, self.reg, self.cl_rate, self.eps, self.tau = n_layers, lr, reg, cl_rate, eps, tau
        self.loss_eps, self.seed = loss_eps, seed
        rp = self.rp = RowPartition(comm, self.n, self.ld)
        lo, hi, pad = rp.lo, rp.hi, rp.rows_pad
        indptr, indices, values = adj
        blk_ptr = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64)
        blk_ptr = np.concatenate([blk_ptr, np.full(pad - (hi - lo), blk_ptr[-1] if hi > lo else 0, np.int64)])   # pad rows: empty
        sel = slice(int(indptr[lo]), int(indptr[hi]))
        self.plan = SpmmPlan(blk_ptr, indices[sel], values[sel], self.ld)     # rows = this rank's block, columns global
        _setup_row_exchange(self, adj, blk_ptr, sel)
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        blk = np.zeros((pad, self.ld), np.float32); blk[:hi - lo] = E0[lo:hi]
        self.E = DeviceBuffer.from_numpy(blk)
        zb = lambda: DeviceBuffer.zeros((pad, self.ld), np.float32)
        self.m, self.v = zb(), zb()
        self.Sm, self.S1, self.S2, self.A, self.B = zb(), zb(), zb(), zb(), zb()
        self.V = [[zb(), zb()], [zb(), zb()]]
        full = lambda: DeviceBuffer.zeros((rp.world * pad, self.ld), np.float32)
        self.X_full, self.Sm_full, self.S1_full, self.S2_full, self.dOut_full = full(), full(), full(), full(), full()
        self.d_loss = DeviceBuffer.zeros(2, np.float64)         # [rec, cl (unscaled)]
        self.batch_rows, self._cmp = bool(batch_rows), None
        self.max_unique = max_unique
        self.ws = DeviceBuffer(capi.info_nce_workspace_bytes(max_unique, self.ld), np.uint8)
        f = np.float32
        self.b1, self.b2, self.adam_eps = f(0.9), f(0.999), f(1e-8)
        self.b1p, self.b2p = self.b1, self.b2
        self.step_no = 0

    def _product(self, x, y, stream, accum=None, addend=None):
        """y = the rank's rows of A_hat gather(x) (+ addend); accum += y"""
        plan, X = _row_operand(self, x, stream)
        capi.spmm_csr(plan, X, y, self.ld, d_addend=addend, addend_scale=1.0 if addend is not None else 0.0,
                      d_accum=accum, stream=stream)

    def _encode_three(self, noises, stream):
        """``noises``: optional 2L buffers [pad][ld], THIS RANK'S rows of the injected uniforms (tests)"""
        pad, lo = self.rp.rows_pad, self.rp.lo
        nz = lambda v, k: None if noises is None else noises[(v - 1) * self.L + k]
        sid = lambda v, k: (self.step_no * 2 + (v - 1)) * 64 + k
        sums = [self.Sm, self.S1, self.S2]
        x = None
        for k in range(self.L):
            y0 = self.A if k % 2 == 0 else self.B
            if k == 0:
                self._product(self.E, y0, stream)
                y1, y2 = self.V[0][0], self.V[1][0]
                capi.perturb_two_views(y0, y1, y2, pad, self.d, self.ld, self.eps, nz(1, 0), nz(2, 0), self.seed, sid(1, 0), sid(2, 0),
                                       self.S1, self.S2, self.Sm, stream, philox_row0=lo)
                x = [y0, y1, y2]
                continue
            self._product(x[0], y0, stream, accum=self.Sm)
            for v in (1, 2):
                yv = self.V[v - 1][k % 2]
                self._product(x[v], yv, stream)
                capi.perturb_rows(yv, pad, self.d, self.ld, self.eps, nz(v, k), self.seed, sid(v, k), d_accum=sums[v], stream=stream,
                                  philox_row0=lo)
                x[v] = yv
            x[0] = y0

    def train_step_async(self, d_u, d_i, d_j, B: int, d_uniq_users, n_uu: int, d_uniq_items, n_ui: int, noises=None, stream=None):
        """as SimGCLTrainer.train_step_async; the row ids are ids of the WHOLE tables"""
        if max(n_uu, n_ui) > self.max_unique:
            raise ValueError("more unique rows in the batch than the InfoNCE workspace holds")
        rp, ld, L = self.rp, self.ld, float(self.L)
        self._encode_three(noises, stream)
        cl = self.d_loss.ptr + 8
        whole_first = getattr(self, "batch_rows", True) and B > 0
        if whole_first:
            # round 4 -- the losses read the three layer sums at the batch's rows only: Sm at {u, nu + i, nu + j} (3B rows), S1 / S2 at the batch's
            # unique users and items.  Every rank contributes its rows of those (zeros elsewhere) into ONE packed buffer
            # [Sm: 3B | S1: n_uu + n_ui | S2: n_uu + n_ui] x ld, ONE all-reduce makes it whole everywhere (3.4 MB at B = 2048 instead of three
            # all-gathers of N x ld = 53 MB), the losses run on it with the rows addressed by position, and every rank -- holding the whole
            # batch gradient -- scatters it into a whole-height operand that the first backward product reads without any exchange
            M = self.max_unique
            if getattr(self, "_cmp", None) is None or self._cmp["B"] != B:
                ar = np.arange(max(3 * B, 2 * M), dtype=np.int32)
                self._cmp = dict(B=B, S=DeviceBuffer.zeros((3 * B + 4 * M, ld), np.float32), dSm=DeviceBuffer.zeros((3 * B, ld), np.float32),
                                 dC=DeviceBuffer.zeros((2 * M, ld), np.float32), ar=DeviceBuffer.from_numpy(ar), arB=DeviceBuffer.from_numpy(ar[:B] + B))
            c = self._cmp
            row = ld * 4
            nC = n_uu + n_ui
            pS1, pS2 = c["S"].ptr + 3 * B * row, c["S"].ptr + (3 * B + nC) * row
            capi.batch_rows_gather(self.Sm, ld, rp.lo, rp.hi, d_u, d_i, d_j, B, self.nu, c["S"], stream)
            for blk, base in ((self.S1, pS1), (self.S2, pS2)):
                capi.rows_gather_owned(blk, ld, rp.lo, rp.hi, d_uniq_users, n_uu, base, stream)
                capi.rows_gather_owned(blk, ld, rp.lo, rp.hi, d_uniq_items, n_ui, base + n_uu * row, stream)
            self.comm.allreduce(c["S"], (3 * B + 2 * nC) * ld, capi.F32, stream)
            c["dSm"].fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
            capi.memset(c["dC"].ptr, 0, max(nC, 1) * row, stream)
            capi.bpr_batch_loss_grad(c["S"], L, B, 3 * B, ld, c["ar"], c["ar"], c["arB"], B, self.loss_eps, self.reg, c["dSm"], self.d_loss, stream, ordered=self.ows)
            capi.info_nce_loss_grad(pS1, pS2, L, c["ar"], n_uu, ld, self.tau, self.cl_rate, self.ws, c["dC"], cl, stream)
            capi.info_nce_loss_grad(pS1, pS2, L, c["ar"].ptr + 4 * n_uu, n_ui, ld, self.tau, self.cl_rate, self.ws, c["dC"], cl, stream)
            self.dOut_full.fill_bytes(0, stream)
            capi.batch_rows_scatter_add(self.dOut_full, ld, 0, self.n, d_u, d_i, d_j, B, self.nu, c["dSm"], stream)
            capi.rows_scatter_add_owned(self.dOut_full, ld, 0, self.n, d_uniq_users, n_uu, c["dC"], stream)
            capi.rows_scatter_add_owned(self.dOut_full, ld, 0, self.n, d_uniq_items, n_ui, c["dC"].ptr + n_uu * row, stream)
        else:
            for blk, full in ((self.Sm, self.Sm_full), (self.S1, self.S1_full), (self.S2, self.S2_full)):
                rp.gather_operand(blk, full, stream)
            self.dOut_full.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
            rows_full = rp.world * rp.rows_pad
            if B:
                capi.bpr_batch_loss_grad(self.Sm_full, L, self.nu, rows_full, ld, d_u, d_i, d_j, B, self.loss_eps, self.reg, self.dOut_full,
                                         self.d_loss, stream, ordered=self.ows)
            capi.info_nce_loss_grad(self.S1_full, self.S2_full, L, d_uniq_users, n_uu, ld, self.tau, self.cl_rate, self.ws, self.dOut_full, cl, stream)
            capi.info_nce_loss_grad(self.S1_full, self.S2_full, L, d_uniq_items, n_ui, ld, self.tau, self.cl_rate, self.ws, self.dOut_full, cl, stream)
        dOut_blk = self.dOut_full.ptr + 4 * rp.lo * ld                  # this rank's rows of the output gradient, in place

        def product(x, y, addend, first):
            if first and whole_first:       # the operand is the whole batch gradient, already on this rank
                capi.spmm_csr(self.plan, self.dOut_full, y, ld, d_addend=addend, addend_scale=1.0 if addend is not None else 0.0, stream=stream)
            else:
                self._product(x, y, stream, addend=addend)
        x = dOut_blk
        for k in range(self.L - 1):
            y = self.A if k % 2 == 0 else self.B
            product(x, y, dOut_blk, k == 0)
            x = y
        g = self.B if x is self.A else self.A
        product(x, g, None, self.L == 1)
        f = np.float32
        alpha = float(f(f(self.lr) * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p)))
        capi.adam_step(self.E, self.m, self.v, g, (rp.hi - rp.lo) * ld, 1.0 / L, alpha, float(self.b1), float(self.b2), float(self.adam_eps), stream)
        self.b1p = f(self.b1p * self.b1); self.b2p = f(self.b2p * self.b2)
        self.step_no += 1

    def losses(self, stream=None):
        rec, cl = self.d_loss.numpy(stream)
        cl *= self.cl_rate
        return float(rec + cl), float(rec), float(cl)

    def main_embeddings(self):
        """(U, V) of the clean encoder (SimGCL.py:23-27), whole, on every rank"""
        x = self.E
        self.Sm.fill_bytes(0)
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            self._product(x, y, None, accum=self.Sm)
            x = y
        self.rp.gather_operand(self.Sm, self.Sm_full)
        m = (self.Sm_full.numpy()[:self.n, :self.d] / np.float32(self.L)).astype(np.float32)
        return np.ascontiguousarray(m[:self.nu]), np.ascontiguousarray(m[self.nu:])

    def block(self, buf) -> np.ndarray:
        return buf.numpy()[:self.rp.hi - self.rp.lo, :self.d].copy()


class _Adam:
    """host-side bookkeeping of one TF-1.14 Adam slot set (fp32 beta powers)"""

    def __init__(self, theta: DeviceBuffer, lr: float):
        self.theta = theta
        self.m = DeviceBuffer.zeros(theta.shape, np.float32); self.v = DeviceBuffer.zeros(theta.shape, np.float32)
        f = np.float32
        self.lr, self.b1, self.b2, self.eps = f(lr), f(0.9), f(0.999), f(1e-8)
        self.b1p, self.b2p = self.b1, self.b2
        self.n = int(np.prod(theta.shape))

    def step(self, grad, grad_scale=1.0, stream=None, grad_l2=0.0):
        f = np.float32
        alpha = float(f(self.lr * np.sqrt(f(1) - self.b2p, dtype=f) / (f(1) - self.b1p)))
        capi.adam_step(self.theta, self.m, self.v, grad, self.n, grad_scale, alpha, float(self.b1), float(self.b2),
                       float(self.eps), stream, grad_l2=grad_l2)
        self.b1p = f(self.b1p * self.b1); self.b2p = f(self.b2p * self.b2)
        self.last = (grad, grad_scale, grad_l2)

    def applied_gradient(self, before=None):
        return _applied_gradient(self.last, before)


class NGCFTrainer:
    """model/ranking/NGCF.py:9-63 on the device: two layers of
    side = A E; E' = dropout(leaky_relu((side+E) W1 + (E*side) W2)); out = [E0 | norm(E1) | norm(E2)],
    batch BPR loss + batch L2 on the 3d-wide rows, Adam on U, V and the four d x d weights."""

    KEEP = 0.9
    N_LAYERS = 2

    def __init__(self, U0, V0, W, adj, lr: float, reg: float, loss_eps: float = 1e-7, seed: int = 0):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        if 3 * self.d > 256:
            raise ValueError("NGCF on the device supports embedding sizes up to 85 (3d <= 256)")
        self.ld = padded_ld(self.d, np.float32)
        self.wide_d, self.wide_ld = 3 * self.d, padded_ld(3 * self.d, np.float32)
        self.lr, self.reg, self.loss_eps, self.seed = lr, reg, loss_eps, seed
        self.plan = SpmmPlan(adj[0], adj[1], adj[2], self.ld, split_row=self.nu)
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.E = [DeviceBuffer.from_numpy(E0), z(), z()]          # E_0 (parameters), E_1, E_2
        self.side = [z(), z()]; self.gate = [z(), z()]
        self.inv = [DeviceBuffer.zeros(self.n, np.float32), DeviceBuffer.zeros(self.n, np.float32)]
        self.dpre, self.dside, self.dEa, self.dEb = z(), z(), z(), z()
        self.All = DeviceBuffer.zeros((self.n, self.wide_ld), np.float32)
        self.dAll = DeviceBuffer.zeros((self.n, self.wide_ld), np.float32)
        pad = lambda w: np.pad(np.asarray(w, np.float32), ((0, self.ld - self.d), (0, self.ld - self.d)))
        # the four d x d weights (and their gradients) are windows of ONE buffer each: one Adam launch updates all four
        ll = self.ld * self.ld
        self.W_all = DeviceBuffer.from_numpy(np.stack([pad(w) for pair in W for w in pair]))
        self.gW_all = DeviceBuffer.zeros((4, self.ld, self.ld), np.float32)
        self.W = [[capi.DeviceSlice(self.W_all, (2 * k + t) * ll, (self.ld, self.ld)) for t in range(2)] for k in range(2)]
        self.gW = [[capi.DeviceSlice(self.gW_all, (2 * k + t) * ll, (self.ld, self.ld)) for t in range(2)] for k in range(2)]
        self.partial = DeviceBuffer(capi.ngcf_wgrad_partial_bytes(self.n, self.ld), np.uint8)
        self.optE = _Adam(self.E[0], lr)
        self.optW = _Adam(self.W_all, lr)
        self.d_loss = DeviceBuffer.zeros(1, np.float64)
        self.row_mask = DeviceBuffer.zeros((self.n + 31) // 32, np.uint32)
        self.batch_rows = None  # capi.RowSubset of the step's rows (ld <= 64: the last layer runs on these rows only)
        self.step_no = 0
        self.dp = None          # dist.BatchParallel

    def forward(self, training: bool, masks=None, stream=None, last_rows=None, last_subset=None):
        """fills E_1, E_2, side, gate, inv and the wide table All = [E_0 | z_1 | z_2].  ``last_rows`` (training):
        the last layer's neighbourhood sum is only formed at the batch's rows -- its output block z_2 is read there
        and nowhere else, and its backward only touches rows with a non-zero gradient, which are the same rows
        (the other rows of side/gate/E_2 keep older, finite values whose gradient weight is exactly 0).
        ``last_subset`` (the same rows as a list): the last layer's dense product and activation also run on those
        rows only -- about 6 k of 70 k rows at the reference's batch size."""
        n, d, ld = self.n, self.d, self.ld
        capi.copy_cols(self.All, self.wide_ld, self.E[0], ld, 0, n, d, False, stream, rows=last_subset)   # the ego block: read where the batch looks
        for k in range(self.N_LAYERS):
            last = k == self.N_LAYERS - 1
            rows = last_subset if last else None
            capi.spmm_csr(self.plan, self.E[k], self.side[k], ld, stream=stream, d_y_row_mask=last_rows if last else None)
            capi.ngcf_dense_fwd(self.E[k], self.side[k], self.W[k][0], self.W[k][1], n, ld, self.gate[k], stream, rows=rows)
            capi.ngcf_activate(self.gate[k], n, d, ld, self.KEEP if training else 1.0,
                               None if masks is None else masks[k], self.seed, self.step_no * 8 + k, self.E[k + 1],
                               self.All, self.wide_ld, (k + 1) * d, self.inv[k], stream, rows=rows)

    def train_step_async(self, d_u, d_i, d_j, B: int, masks=None, stream=None):
        n, d, ld = self.n, self.d, self.ld
        subset = None
        if B and ld <= 64:
            bound = min(3 * B, n)
            if self.batch_rows is None or self.batch_rows.capacity < bound:
                self.batch_rows = capi.RowSubset(bound)
            subset = capi.mark_compact_batch_rows(d_u, d_i, d_j, B, self.nu, n, self.row_mask, self.batch_rows, bound, stream,
                                                  d_zero8=self.d_loss, n_zero8=1)
        else:
            self.row_mask.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
            if B:
                capi.mark_batch_rows(d_u, d_i, d_j, B, self.nu, self.row_mask, stream)
        self.forward(True, masks, stream, last_rows=self.row_mask, last_subset=subset)
        # the loss scatters into the batch's rows of dAll and nothing reads another row of it (wide_mask below): clearing
        # those ~6 k rows replaces a 71 MB fill of the wide table
        wide_mask = self.row_mask if subset is not None else None
        if subset is not None:
            capi.zero_rows(self.dAll, self.wide_ld, subset, stream)
        else:
            self.dAll.fill_bytes(0, stream)
        if B:
            capi.bpr_batch_loss_grad(self.All, 1.0, self.nu, n, self.wide_ld, d_u, d_i, d_j, B, self.loss_eps, self.reg,
                                     self.dAll, self.d_loss, stream, d_row_mask=self.row_mask, ordered=self.ows)
        dnext = None
        for k in (1, 0):
            dE = self.dEa if k == 1 else self.dEb
            rows = subset if k == self.N_LAYERS - 1 else None
            capi.ngcf_layer_bwd(dnext, self.dAll, self.All, self.wide_ld, (k + 1) * d, self.inv[k], self.gate[k], self.E[k],
                                self.side[k], self.W[k][0], self.W[k][1], n, d, ld, self.dpre, self.dside, dE, self.partial,
                                self.gW[k][0], self.gW[k][1], stream, rows=rows, d_wide_row_mask=wide_mask)
            # dE += A^T dside.  For the last layer dside is non-zero only on the batch rows (its gradient
            # comes from the concat block alone), so that SpMM skips the other operand rows.
            capi.spmm_csr(self.plan, self.dside, dE, ld, d_addend=dE, addend_scale=1.0, stream=stream,
                          d_x_row_mask=self.row_mask if k == 1 else None,
                          d_addend_row_mask=self.row_mask if rows is not None else None)   # last layer: dE holds the batch's rows only
            dnext = dE
        capi.copy_cols(dnext, ld, self.dAll, self.wide_ld, 0, n, d, True, stream, rows=subset)          # + ego block of the concat
        if self.dp is not None:     # table gradient + the four d x d weight gradients ("all-reduce for the dense layers")
            self.dp.all_reduce(dnext); self.dp.all_reduce(self.d_loss)
            self.dp.all_reduce(self.gW_all)
        self.optE.step(dnext, stream=stream)
        self.optW.step(self.gW_all, stream=stream)
        self.step_no += 1

    def loss(self, stream=None) -> float:
        return float(self.d_loss.numpy(stream)[0])

    def inference_embeddings(self):
        """3d-wide (U, V) of the inference graph (isTraining = 0, NGCF.py:65-69)."""
        self.forward(False)
        A = self.All.numpy()[:, :self.wide_d]
        return np.ascontiguousarray(A[:self.nu]), np.ascontiguousarray(A[self.nu:])

    def parameters(self):
        E = self.E[0].numpy()[:, :self.d]
        return E[:self.nu].copy(), E[self.nu:].copy(), [[w.numpy()[:self.d, :self.d].copy() for w in pair] for pair in self.W]

    def gradients(self):
        """(dU, dV, [[dW1, dW2] per layer]) of the last step, before Adam"""
        g = self.optE.applied_gradient()[:, :self.d]
        gw = self.optW.applied_gradient()
        return g[:self.nu], g[self.nu:], [[gw[2 * k + t, :self.d, :self.d] for t in range(2)] for k in range(2)]


class RowPartitionedNGCFTrainer:
    """NGCF with every node table ROW-PARTITIONED over the ranks (BASELINE config #5: "row-shard the tables,
    all-reduce for the dense layers").  Rank r owns rows [lo, hi) of E_0 and its Adam slots, of the adjacency, and of every
    per-layer table (side, gate, E_k, 1/|.|); the four d x d weights and their Adam slots are replicated.  One step at
    the reference's batch size, the same step as ``NGCFTrainer``:

        forward   per layer: all-gather of the E_k blocks -> side = the rank's rows of A_hat E_k -> the dense product and
                  the activation on the rank's rows (dropout keyed by the TABLE row, so the same entries drop as on one GPU);
                  the normalised block z_k is all-gathered into every rank's copy of the wide table [E_0 | z_1 | z_2];
        loss      every rank evaluates the whole batch on its copy (B rows: ~2 % of a step) and keeps its rows of the gradient;
        backward  per layer on the rank's rows: dpre, the two MFMA products, the weight gradients summed over the rank's rows;
                  dE += A_hat dside is again all-gather + the rank's rows (A_hat is symmetric);
        update    ONE all-reduce of the four weight gradients ("all-reduce for the dense layers"), Adam on the replicated
                  weights (identical on every rank) and on the rank's rows of E_0.

    Per step and rank: 2L all-gathers of N x ld floats for the propagation, L for the z blocks (17.8 MB each at the
    Yelp2018 shape) and a 64 KB all-reduce.  No rank holds more than its rows of the parameters and per-layer tables;
    the gathered operand and the wide table are transients of the step."""

    KEEP = 0.9
    N_LAYERS = 2

    def __init__(self, comm, U0, V0, W, adj, lr: float, reg: float, loss_eps: float = 1e-7, seed: int = 0, batch_rows: bool = True):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        from .dist import RowPartition
        self.comm = comm
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        if 3 * self.d > 256:
            raise ValueError("NGCF on the device supports embedding sizes up to 85 (3d <= 256)")
        self.ld = padded_ld(self.d, np.float32)
        self.wide_d, self.wide_ld = 3 * self.d, padded_ld(3 * self.d, np.float32)
        self.lr, self.reg, self.loss_eps, self.seed = lr, reg, loss_eps, seed
        self.batch_rows, self._batch = bool(batch_rows), {}
        rp = self.rp = RowPartition(comm, self.n, self.ld)
        lo, hi, pad = rp.lo, rp.hi, rp.rows_pad
        indptr, indices, values = adj
        blk_ptr = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64)
        blk_ptr = np.concatenate([blk_ptr, np.full(pad - (hi - lo), blk_ptr[-1] if hi > lo else 0, np.int64)])   # pad rows: empty
        sel = slice(int(indptr[lo]), int(indptr[hi]))
        self.plan = SpmmPlan(blk_ptr, indices[sel], values[sel], self.ld)     # rows = this rank's block, columns global
        _setup_row_exchange(self, adj, blk_ptr, sel)
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        blk = np.zeros((pad, self.ld), np.float32); blk[:hi - lo] = E0[lo:hi]
        zb = lambda: DeviceBuffer.zeros((pad, self.ld), np.float32)
        self.E = [DeviceBuffer.from_numpy(blk), zb(), zb()]
        self.side = [zb(), zb()]; self.gate = [zb(), zb()]; self.z = [zb(), zb()]
        self.inv = [DeviceBuffer.zeros(pad, np.float32), DeviceBuffer.zeros(pad, np.float32)]
        self.dpre, self.dside, self.dEa, self.dEb = zb(), zb(), zb(), zb()
        rows_full = rp.world * pad
        self.X_full = DeviceBuffer.zeros((rows_full, self.ld), np.float32)
        self.All_full = DeviceBuffer.zeros((rows_full, self.wide_ld), np.float32)
        self.dAll_full = DeviceBuffer.zeros((rows_full, self.wide_ld), np.float32)
        ll = self.ld * self.ld
        padw = lambda w: np.pad(np.asarray(w, np.float32), ((0, self.ld - self.d), (0, self.ld - self.d)))
        self.W_all = DeviceBuffer.from_numpy(np.stack([padw(w) for pair in W for w in pair]))
        self.gW_all = DeviceBuffer.zeros((4, self.ld, self.ld), np.float32)
        self.W = [[capi.DeviceSlice(self.W_all, (2 * k + t) * ll, (self.ld, self.ld)) for t in range(2)] for k in range(2)]
        self.gW = [[capi.DeviceSlice(self.gW_all, (2 * k + t) * ll, (self.ld, self.ld)) for t in range(2)] for k in range(2)]
        self.partial = DeviceBuffer(capi.ngcf_wgrad_partial_bytes(pad, self.ld), np.uint8)
        self.optE, self.optW = _Adam(self.E[0], lr), _Adam(self.W_all, lr)
        self.d_loss = DeviceBuffer.zeros(1, np.float64)
        self.step_no = 0

    def forward(self, training: bool, masks=None, stream=None, own_rows_only: bool = False):
        """E_1, E_2, side, gate, inv on the rank's rows; the wide table All = [E_0 | z_1 | z_2] whole on every rank -- or, with
        ``own_rows_only`` (the training step, round 4), only this rank's rows of it: the loss needs the batch's rows, which travel
        separately (``train_step_async``), not two all-gathers of N x ld floats.
        ``masks``: per layer, THIS RANK'S rows [pad][ld] of the injected keep decisions (tests)."""
        rp, d, ld, pad = self.rp, self.d, self.ld, self.rp.rows_pad
        rows_full = rp.world * pad
        mine = 4 * rp.lo * self.wide_ld
        for k in range(self.N_LAYERS):
            if k == 0:      # the ego block of the concat is the WHOLE E_0 (the loss looks up any row): this product's operand is gathered whole
                rp.gather_operand(self.E[k], self.X_full, stream)
                capi.copy_cols(self.All_full, self.wide_ld, self.X_full, ld, 0, rows_full, d, False, stream)
                plan, X = self.plan, self.X_full
            else:
                plan, X = _row_operand(self, self.E[k], stream)
            capi.spmm_csr(plan, X, self.side[k], ld, stream=stream)
            capi.ngcf_dense_fwd(self.E[k], self.side[k], self.W[k][0], self.W[k][1], pad, ld, self.gate[k], stream)
            capi.ngcf_activate(self.gate[k], pad, d, ld, self.KEEP if training else 1.0, None if masks is None else masks[k],
                               self.seed, self.step_no * 8 + k, self.E[k + 1], self.z[k], ld, 0, self.inv[k], stream,
                               philox_row0=rp.lo)
            if own_rows_only:
                capi.copy_cols(self.All_full.ptr + mine + 4 * (k + 1) * d, self.wide_ld, self.z[k], ld, 0, pad, d, False, stream)
            else:
                rp.gather_operand(self.z[k], self.X_full, stream)
                capi.copy_cols(self.All_full.ptr + 4 * (k + 1) * d, self.wide_ld, self.X_full, ld, 0, rows_full, d, False, stream)

    def _batch_buffers(self, B: int):
        b = self._batch.get(B)
        if b is None:
            ar = np.arange(max(B, 1), dtype=np.int32)
            b = self._batch[B] = dict(S=DeviceBuffer.zeros((3 * max(B, 1), self.wide_ld), np.float32), dS=DeviceBuffer.zeros((3 * max(B, 1), self.wide_ld), np.float32),
                                      u=DeviceBuffer.from_numpy(ar), j=DeviceBuffer.from_numpy(ar + B))
        return b

    def train_step_async(self, d_u, d_i, d_j, B: int, masks=None, stream=None):
        rp, d, ld, pad = self.rp, self.d, self.ld, self.rp.rows_pad
        rows_full = rp.world * pad
        mine = 4 * rp.lo * self.wide_ld                 # byte offset of this rank's rows in the wide tables
        if self.batch_rows and B:
            # round 4: the two all-gathers of the normalised blocks z_k are gone -- every rank contributes its rows of the batch's 3B rows of
            # the wide table (zeros elsewhere), ONE all-reduce of 3B x wide_ld floats makes them whole everywhere, every rank evaluates the
            # whole batch on that compact table and scatters its own rows of the gradient back into its block
            self.forward(True, masks, stream, own_rows_only=True)
            b = self._batch_buffers(B)
            capi.batch_rows_gather(self.All_full.ptr + mine, self.wide_ld, rp.lo, rp.hi, d_u, d_i, d_j, B, self.nu, b["S"], stream)
            self.comm.allreduce(b["S"], 3 * B * self.wide_ld, capi.F32, stream)
            b["dS"].fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
            capi.bpr_batch_loss_grad(b["S"], 1.0, B, 3 * B, self.wide_ld, b["u"], b["u"], b["j"], B, self.loss_eps, self.reg, b["dS"], self.d_loss, stream, ordered=self.ows)
            capi.memset(self.dAll_full.ptr + mine, 0, pad * self.wide_ld * 4, stream)
            capi.batch_rows_scatter_add(self.dAll_full.ptr + mine, self.wide_ld, rp.lo, rp.hi, d_u, d_i, d_j, B, self.nu, b["dS"], stream)
        else:
            self.forward(True, masks, stream)
            self.dAll_full.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
            if B:
                capi.bpr_batch_loss_grad(self.All_full, 1.0, self.nu, rows_full, self.wide_ld, d_u, d_i, d_j, B, self.loss_eps, self.reg,
                                         self.dAll_full, self.d_loss, stream, ordered=self.ows)
        dnext = None
        for k in (1, 0):
            dE = self.dEa if k == 1 else self.dEb
            capi.ngcf_layer_bwd(dnext, self.dAll_full.ptr + mine, self.All_full.ptr + mine, self.wide_ld, (k + 1) * d, self.inv[k],
                                self.gate[k], self.E[k], self.side[k], self.W[k][0], self.W[k][1], pad, d, ld, self.dpre, self.dside, dE,
                                self.partial, self.gW[k][0], self.gW[k][1], stream)
            plan, X = _row_operand(self, self.dside, stream)             # dE += (A_hat dside)[lo:hi]
            capi.spmm_csr(plan, X, dE, ld, d_addend=dE, addend_scale=1.0, stream=stream)
            dnext = dE
        capi.copy_cols(dnext, ld, self.dAll_full.ptr + mine, self.wide_ld, 0, pad, d, True, stream)   # + ego block of the concat
        self.comm.allreduce(self.gW_all, 4 * ld * ld, capi.F32, stream)  # each rank summed its own rows: the dense layers' all-reduce
        self.optE.step(dnext, stream=stream)
        self.optW.step(self.gW_all, stream=stream)
        self.step_no += 1

    def loss(self, stream=None) -> float:
        return float(self.d_loss.numpy(stream)[0])

    def inference_embeddings(self):
        """3d-wide (U, V) of the inference graph (isTraining = 0, NGCF.py:65-69), whole, on every rank"""
        self.forward(False)
        A = self.All_full.numpy()[:self.n, :self.wide_d]
        return np.ascontiguousarray(A[:self.nu]), np.ascontiguousarray(A[self.nu:])

    def block(self, buf) -> np.ndarray:
        """this rank's rows of a block buffer (pad rows and pad columns dropped)"""
        return buf.numpy()[:self.rp.hi - self.rp.lo, :self.d].copy()

    def weights(self):
        return [[w.numpy()[:self.d, :self.d].copy() for w in pair] for pair in self.W]


class SGLTrainer:
    """model/ranking/SGL.py on the device: the recommendation view (LightGCN over the full adjacency) plus
    two views over per-epoch augmented sub-graphs, BPR on the main view, InfoNCE (users and items of the
    batch merged into one contrast set, SGL.py:192-217) between the sub-graph views, Adam.
    ``set_subgraphs`` installs the epoch's sub-adjacencies: one CSR per view (node / edge dropout) or one
    per view and layer (random walk)."""

    def __init__(self, U0, V0, adj, n_layers: int, lr: float, reg: float, ssl_reg: float, temp: float,
                 loss_eps: float = 1e-7, max_unique: int = 8192):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.L, self.reg, self.ssl_reg, self.temp, self.loss_eps = n_layers, reg, ssl_reg, temp, loss_eps
        self.main_plan = SpmmPlan(adj[0], adj[1], adj[2], self.ld, split_row=self.nu)
        self.plans = [None, None]                 # per view: list of L plans
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        self.E = DeviceBuffer.from_numpy(E0)
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.S = [z(), z(), z()]                  # layer sums of main, view 1, view 2
        self.dOut = [z(), z(), z()]
        self.G, self.A, self.B = z(), z(), z()
        self.opt = _Adam(self.E, lr)
        self.d_loss = DeviceBuffer.zeros(2, np.float64)     # [rec, ssl (unscaled)]
        self.row_mask = DeviceBuffer.zeros((self.n + 31) // 32, np.uint32)
        self.max_unique = max_unique
        self.ws = DeviceBuffer(capi.info_nce_workspace_bytes(max_unique, self.ld), np.uint8)

    def set_subgraphs(self, adjs1, adjs2):
        """adjs*: one (indptr, indices, values) triple, or a list of L of them (random walk)."""
        def plans(adjs):
            if isinstance(adjs, tuple):
                p = SpmmPlan(adjs[0], adjs[1], adjs[2], self.ld, split_row=self.nu, row_chunk=self.main_plan.row_chunk)
                return [p] * self.L
            assert len(adjs) == self.L
            return [SpmmPlan(a[0], a[1], a[2], self.ld, split_row=self.nu, row_chunk=self.main_plan.row_chunk) for a in adjs]
        self.plans = [plans(adjs1), plans(adjs2)]

    def set_subgraph_values(self, vals1, vals2):
        """device-drawn sub-graphs (SubgraphSampler.draw over THIS trainer's full graph): one value array per view, or a list of L"""
        def plans(vals):
            if isinstance(vals, DeviceBuffer):
                return [self.main_plan.with_values(vals)] * self.L
            assert len(vals) == self.L
            return [self.main_plan.with_values(v) for v in vals]
        self.plans = [plans(vals1), plans(vals2)]

    def _view_plans(self, v):
        return [self.main_plan] * self.L if v == 0 else self.plans[v - 1]

    def _forward(self, v, stream=None, last_rows=None):
        S = self.S[v]
        x = self.E
        plans = self._view_plans(v)
        for k, plan in enumerate(plans):
            y = self.A if k % 2 == 0 else self.B
            capi.spmm_csr(plan, x, y, self.ld, d_accum=S, stream=stream, d_accum_init=self.E if k == 0 else None,   # S = E + M_1 E: no copy
                          d_y_row_mask=last_rows if k == len(plans) - 1 else None)   # last layer: batch rows only
            x = y

    def _backward(self, v, stream=None):
        """G += sum_k (prod of the view's matrices)^T applied to dOut[v]  (times 1/(L+1) in Adam)."""
        d = self.dOut[v]
        plans = self._view_plans(v)
        if self.L == 0:
            raise ValueError("SGL needs at least one layer")
        x = d
        for step, k in enumerate(range(self.L - 1, -1, -1)):      # dE_k = dOut + M_k^T dE_{k+1}; matrices are symmetric
            last = (k == 0)
            y = self.A if step % 2 == 0 else self.B
            capi.spmm_csr(plans[k], x, y, self.ld, d_addend=d, addend_scale=1.0, d_accum=self.G if last else None,
                          stream=stream, d_x_row_mask=self.row_mask if step == 0 else None, d_addend_row_mask=self.row_mask)
            x = y

    def train_step_async(self, d_u, d_i, d_j, B: int, d_rows, n_rows: int, stream=None, share=None):
        """d_rows: the batch's unique users followed by its unique positive items (+n_users), distinct.
        ``share`` = (offset, count), one process per GPU (dist.BatchParallel in ``self.dp``): the BPR term covers this
        rank's rows of the step, the InfoNCE term (which couples all the step's rows) is formed whole with weight
        ssl_reg / world, and the table gradient is summed over the ranks before Adam."""
        if self.plans[0] is None:
            raise RuntimeError("set_subgraphs() first")
        if n_rows > self.max_unique:
            raise ValueError("more unique rows in the batch than the InfoNCE workspace holds")
        div = float(self.L + 1)
        dp = getattr(self, "dp", None)
        lo, cnt = (0, B) if share is None else share
        # the batch's rows as bitmap + list (one launch, loss accumulators cleared with it): the three output gradients are written
        # and read at those rows only (operand / addend masks of the backward products), so only those rows are cleared
        bound = min(3 * B, self.n)
        if getattr(self, "batch_rows", None) is None or self.batch_rows.capacity < bound:
            self.batch_rows = capi.RowSubset(bound)
        subset = capi.mark_compact_batch_rows(d_u, d_i, d_j, B, self.nu, self.n, self.row_mask, self.batch_rows, bound, stream,
                                              d_zero8=self.d_loss, n_zero8=2)
        for v in range(3):
            self._forward(v, stream, last_rows=self.row_mask)
            capi.zero_rows(self.dOut[v], self.ld, subset, stream)
        self.G.fill_bytes(0, stream)
        if share is not None:
            d_u, d_i, d_j = (capi.device_ptr(p) + 4 * lo for p in (d_u, d_i, d_j))
        if cnt:
            capi.bpr_batch_loss_grad(self.S[0], div, self.nu, self.n, self.ld, d_u, d_i, d_j, cnt, self.loss_eps, self.reg,
                                     self.dOut[0], self.d_loss, stream, d_row_mask=self.row_mask, ordered=self.ows)
        capi.info_nce_loss_grad(self.S[1], self.S[2], div, d_rows, n_rows, self.ld, self.temp,
                                self.ssl_reg if dp is None else self.ssl_reg / dp.world, self.ws,
                                self.dOut[1], self.d_loss.ptr + 8, stream, d_out2=self.dOut[2])
        for v in range(3):
            self._backward(v, stream)
        if dp is not None:
            dp.all_reduce(self.G); dp.all_reduce(self.d_loss.head_view(1))
        self.opt.step(self.G, grad_scale=1.0 / div, stream=stream)

    def losses(self, stream=None):
        rec, ssl = self.d_loss.numpy(stream)
        return float(rec + self.ssl_reg * ssl), float(rec), float(self.ssl_reg * ssl)

    def gradients(self):
        g = self.opt.applied_gradient()[:, :self.d]
        return g[:self.nu], g[self.nu:]

    def main_embeddings(self):
        self._forward(0)
        m = (self.S[0].numpy()[:, :self.d] / np.float32(self.L + 1)).astype(np.float32)
        return np.ascontiguousarray(m[:self.nu]), np.ascontiguousarray(m[self.nu:])

    def ego_embeddings(self):
        E = self.E.numpy()
        return E[:self.nu, :self.d].copy(), E[self.nu:, :self.d].copy()


class BUIRTrainer:
    """model/ranking/BUIR.py:13-172 on the device.  Online encoder = LightGCN mean over this epoch's sub-graph O plus
    q = tanh(. W + b); target encoder = the same propagation of the momentum tables over sub-graph T (no gradient);
    loss = symmetric (1 - cosine) between q of one side and the target of the other; Adam on the online tables, W, b;
    target <- target*tau + online*(1 - tau) after every step.  Only the batch's rows of q are ever looked up, so the
    linear layer runs on <= 2B rows (qrec_buir_batch_loss_grad) and the last propagation layer of both encoders is
    computed at the batch's rows only."""

    def __init__(self, U0, V0, W0, b0, n_layers: int, lr: float, tau: float):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        if self.ld > 128:
            raise ValueError("BUIR kernels support embedding sizes up to 128")
        if n_layers < 1:
            raise ValueError("BUIR needs at least one propagation layer")
        self.L, self.tau = n_layers, float(np.float32(tau))
        E0 = np.zeros((self.n, self.ld), np.float32)
        E0[:self.nu, :self.d] = U0; E0[self.nu:, :self.d] = V0
        self.E, self.T = DeviceBuffer.from_numpy(E0), DeviceBuffer.from_numpy(E0)      # target starts as a copy (BUIR.py:85-86)
        Wp = np.zeros((self.ld, self.ld), np.float32); Wp[:self.d, :self.d] = W0
        bp = np.zeros(self.ld, np.float32); bp[:self.d] = np.asarray(b0, np.float32).reshape(-1)
        self.W, self.b = DeviceBuffer.from_numpy(Wp), DeviceBuffer.from_numpy(bp)
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.S_on, self.S_tar, self.dS, self.A, self.B = z(), z(), z(), z(), z()
        self.gW, self.gb = DeviceBuffer.zeros((self.ld, self.ld), np.float32), DeviceBuffer.zeros(self.ld, np.float32)
        self.wscratch = DeviceBuffer(capi.buir_wgrad_scratch_bytes(self.ld), np.uint8)
        self.optE, self.optW, self.optb = _Adam(self.E, lr), _Adam(self.W, lr), _Adam(self.b, lr)
        self.row_mask = DeviceBuffer.zeros((self.n + 31) // 32, np.uint32)
        self.d_loss = DeviceBuffer.zeros(1, np.float64)
        self.plan_o = self.plan_t = None
        self._cap = 0
        self.Xb = self.Gb = None

    def set_subgraphs(self, adj_o, adj_t):
        """this epoch's two normalized sub-graph adjacencies (CSR triples), BUIR.py:139-146"""
        # the row -> XCD-run map is computed on the first sub-graph and kept: every later epoch's sub-graphs are draws from the same graph
        self.plan_o = SpmmPlan(adj_o[0], adj_o[1], adj_o[2], self.ld, split_row=self.nu, row_chunk=getattr(self, "_row_chunk", None))
        self._row_chunk = self.plan_o.row_chunk
        self.plan_t = SpmmPlan(adj_t[0], adj_t[1], adj_t[2], self.ld, split_row=self.nu, row_chunk=self._row_chunk)

    def set_full_graph(self, adj):
        """throughput mode: the plan of the FULL graph, whose structure every device-drawn sub-graph reuses (set_subgraph_values)"""
        self.full_plan = SpmmPlan(adj[0], adj[1], adj[2], self.ld, split_row=self.nu)
        self._row_chunk = self.full_plan.row_chunk

    def set_subgraph_values(self, vals_o: DeviceBuffer, vals_t: DeviceBuffer):
        self.plan_o, self.plan_t = self.full_plan.with_values(vals_o), self.full_plan.with_values(vals_t)

    def _mean_sum(self, plan, X, S, stream=None, last_rows=None):
        """S = X + A X + ... + A^L X (the mean's 1/(L+1) is applied where S is used)"""
        if self.L == 0:
            S.copy_from(X, stream)
        x = X
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            capi.spmm_csr(plan, x, y, self.ld, d_accum=S, stream=stream, d_accum_init=X if k == 0 else None,    # S = X + A X: no copy
                          d_y_row_mask=last_rows if k == self.L - 1 else None)
            x = y

    def train_step_async(self, d_u, d_i, B: int, stream=None):
        if self.plan_o is None:
            raise RuntimeError("set_subgraphs() first")
        if B > self._cap:
            self.Xb, self.Gb = DeviceBuffer((2 * B, self.ld), np.float32), DeviceBuffer((2 * B, self.ld), np.float32)
            self._cap = B
        div = float(self.L + 1)
        # rows {u, nu+i} of the batch as bitmap + list, loss accumulator cleared by the same launch; dS is written and read at
        # those rows only (operand / addend masks below), so only those rows are cleared
        bound = min(2 * B, self.n)
        if getattr(self, "batch_rows", None) is None or self.batch_rows.capacity < bound:
            self.batch_rows = capi.RowSubset(max(bound, 1))
        subset = capi.mark_compact_batch_rows(d_u, d_i, d_i, B, self.nu, self.n, self.row_mask, self.batch_rows, bound, stream,
                                              d_zero8=self.d_loss, n_zero8=1)
        self._mean_sum(self.plan_o, self.E, self.S_on, stream, self.row_mask)
        self._mean_sum(self.plan_t, self.T, self.S_tar, stream, self.row_mask)
        capi.zero_rows(self.dS, self.ld, subset, stream)
        capi.buir_batch_loss_grad(self.S_on, self.S_tar, div, self.nu, self.ld, self.W, self.b, d_u, d_i, B, self.dS, self.Xb,
                                  self.Gb, self.d_loss, stream, ordered=self.ows)
        if B:
            capi.buir_wgrad(self.Xb, self.Gb, 2 * B, self.ld, self.wscratch, self.gW, self.gb, stream)
        else:                   # an empty share of a step (multi-GPU tail batch)
            self.gW.fill_bytes(0, stream); self.gb.fill_bytes(0, stream)
        # d online tables = (1/(L+1)) (I + A + ... + A^L) dS  (A symmetric): H_0 = dS, H_{k+1} = dS + A H_k
        x = self.dS
        for k in range(self.L):
            y = self.A if k % 2 == 0 else self.B
            capi.spmm_csr(self.plan_o, x, y, self.ld, d_addend=self.dS, addend_scale=1.0, stream=stream,
                          d_x_row_mask=self.row_mask if k == 0 else None, d_addend_row_mask=self.row_mask)
            x = y
        dp = getattr(self, "dp", None)
        if dp is not None:      # every term is a sum over the step's pairs: the ranks' shares add up
            for buf in (x, self.gW, self.gb, self.d_loss):
                dp.all_reduce(buf)
        self.optE.step(x, grad_scale=1.0 / div, stream=stream)
        self.optW.step(self.gW, stream=stream); self.optb.step(self.gb, stream=stream)
        capi.ema_update(self.T, self.E, self.tau, self.n * self.ld, stream)

    def loss(self, stream=None) -> float:
        return float(self.d_loss.numpy(stream)[0])

    def online_tables(self):
        E = self.E.numpy(); return E[:, :self.d].copy()

    def target_tables(self):
        T = self.T.numpy(); return T[:, :self.d].copy()

    def weights(self):
        return self.W.numpy()[:self.d, :self.d].copy(), self.b.numpy()[:self.d].copy()

    def gradients(self):
        """(d online tables [n, d], dW, db) of the last step, before Adam"""
        return (self.optE.applied_gradient()[:, :self.d], self.optW.applied_gradient()[:self.d, :self.d], self.optb.applied_gradient()[:self.d])

    def final_tables(self, adj):
        """(q_user, q_item, o_user, o_item) over the FULL adjacency (BUIR.py:160-167).  The linear layer on all N rows
        runs once per training run: it reuses the batch kernel's forward on the host-side numpy copy."""
        plan = SpmmPlan(adj[0], adj[1], adj[2], self.ld)
        self._mean_sum(plan, self.E, self.S_on)
        online = (self.S_on.numpy()[:, :self.d] / np.float32(self.L + 1)).astype(np.float32)
        W, b = self.weights()
        q = np.tanh(online @ W + b[None, :], dtype=np.float32)
        return q[:self.nu], q[self.nu:], online[:self.nu], online[self.nu:]


# ======================================================================================================================
# SEPT (model/ranking/SEPT.py): social views, the per-epoch perturbed graph, the four-view trainer
# ======================================================================================================================
def _csr_triple(M, dtype=np.float32):
    """(indptr int64, indices int32, values) of a scipy matrix, columns ascending inside each row"""
    M = M.tocsr(); M.sort_indices()
    return M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data.astype(dtype)


def _scaled_by_row_sums(M):
    """diag(s) M diag(s), s = rowsum^-1/2 with inf -> 0, in M's dtype: the ``normalization`` helper of SEPT.py:53-59 and
    the tail of get_adj_mat (:107-113).  ROW sums on both sides even when M is not symmetric."""
    import scipy.sparse as sp
    s = np.asarray(M.sum(axis=1)).ravel()
    with np.errstate(divide="ignore"):
        s = np.power(s, -0.5)
    s[np.isinf(s)] = 0.0
    D = sp.diags(s)
    return D.dot(M).dot(D)


def sept_user_views(n_users: int, n_items: int, uid, iid, follower, followee):
    """The friend view and the sharing view of SEPT (get_birectional_social_matrix, get_social_related_views,
    SEPT.py:42-67) as scipy CSR float64 (the reference builds its float32 tensors from these, :117):
      B       = follow counts squared element-wise (0/1 data: the follow graph itself, directed);
      friend  = (#two-step follow paths u -> w -> v) on the edges u -> v of B, plus the identity;
      sharing = (#items u and v both consumed) on the edges of B, plus the identity;
    both scaled by their row sums on either side."""
    import scipy.sparse as sp
    fo, fe = np.asarray(follower, np.int64), np.asarray(followee, np.int64)
    F = sp.csr_matrix((np.ones(fo.size, np.float32), (fo, fe)), shape=(n_users, n_users))
    B = F.multiply(F).tocsr()
    R = sp.csr_matrix((np.ones(len(uid), np.float32), (np.asarray(uid, np.int64), np.asarray(iid, np.int64))), shape=(n_users, n_items))
    eye = sp.eye(n_users)
    friend = (B @ B).multiply(B) + eye
    sharing = (R @ R.T).multiply(B) + eye
    return _scaled_by_row_sums(friend).tocsr(), _scaled_by_row_sums(sharing).tocsr()


def sept_perturbed_adjacency(state625, n_users: int, n_items: int, uid, iid, follower, followee, drop_rate: float):
    """get_adj_mat(is_subgraph=True) (SEPT.py:79-114): random.sample keeps int(E (1 - rate)) rating edges, then
    int(R (1 - rate)) follow edges (``state625``, the CPython generator, advances in place); the kept rating edges
    go in both ways, the kept follow edges (counts squared) into the user-user block, one way; scaled by row sums.
    float32 throughout, as in the reference.  scipy CSR over the N = n_users + n_items nodes."""
    import scipy.sparse as sp
    n = n_users + n_items
    uid, iid = np.asarray(uid, np.int64), np.asarray(iid, np.int64)
    fo, fe = np.asarray(follower, np.int64), np.asarray(followee, np.int64)
    if drop_rate > 0:
        keep = capi.mt_sample_range(state625, uid.size, int(uid.size * (1 - drop_rate)))
        skeep = capi.mt_sample_range(state625, fo.size, int(fo.size * (1 - drop_rate)))
        up = sp.csr_matrix((np.ones(keep.size, np.float32), (uid[keep], n_users + iid[keep])), shape=(n, n))
        soc = sp.csr_matrix((np.ones(skeep.size, np.float32), (fo[skeep], fe[skeep])), shape=(n, n))
        A = up + up.T + soc.multiply(soc)
    else:
        up = sp.csr_matrix((np.ones(uid.size, np.float32), (uid, n_users + iid)), shape=(n, n))
        A = up + up.T
    return _scaled_by_row_sums(A).tocsr()


class _View:
    """one SEPT view: x_0 = X0, x_k = M x_{k-1}, S = x_0 + sum_k l2_normalize(x_k) over ``rows`` rows; ``planT`` = M^T
    for the backward pass (the friend, sharing and perturbed graphs are not symmetric)."""

    def __init__(self, rows: int, ld: int, L: int):
        self.rows, self.ld, self.L = rows, ld, L
        z = lambda: DeviceBuffer.zeros((max(rows, 1), ld), np.float32)
        self.S, self.dS = z(), z()
        self.x = [z() for _ in range(L)]
        self.inv = [DeviceBuffer.zeros(max(rows, 1), np.float32) for _ in range(L)]
        self.plan = self.planT = None

    def set_matrix(self, M, split_row=None):
        """M: scipy CSR (rows x rows); ``split_row``: see SpmmPlan (joint user-item graphs)"""
        # the row -> XCD-run map of the first matrix is kept for later ones of the same shape (per-epoch perturbed graphs)
        cached = getattr(self, "_row_chunk", None)
        if cached is not None and (split_row is None or cached[0] != (M.shape[0], split_row)):
            cached = None
        self.plan = SpmmPlan(*_csr_triple(M), self.ld, split_row=split_row, row_chunk=None if cached is None else cached[1])
        if split_row is not None and self.plan.row_chunk is not None:
            self._row_chunk = ((M.shape[0], split_row), self.plan.row_chunk)
        self.planT = self.plan if (abs(M - M.T)).nnz == 0 else SpmmPlan(*_csr_triple(M.T), self.ld, split_row=split_row, row_chunk=self.plan.row_chunk)

    def forward(self, X0, stream=None, last_rows=None):
        """``last_rows`` (row bitmap, training): the last layer is only formed at those rows -- S is read at the batch's
        rows and nowhere else, and x_L feeds nothing further (other rows of x_L / S keep stale, finite values)."""
        self.S.copy_from(X0, stream, nbytes=self.rows * self.ld * 4)
        prev = X0
        for k in range(self.L):
            capi.spmm_csr(self.plan, prev, self.x[k], self.ld, stream=stream,
                          d_y_row_mask=last_rows if k == self.L - 1 else None)
            capi.l2norm_rows_accum(self.x[k], self.rows, self.ld, self.S, self.inv[k], stream)
            prev = self.x[k]

    def backward(self, T, G, d_total, stream=None, ds_rows=None):
        """d_total[:rows] += dS + M^T g_1, g_k = normalize_bwd_k(dS) + M^T g_{k+1}  (T, G = [G0, G1]: scratch, rows x ld).
        ``ds_rows``: bitmap of the rows where dS is non-zero; g_L = normalize_bwd_L(dS) is zero elsewhere, so the first
        product skips those operand rows (bit-identical)."""
        g, first = None, True
        for k in range(self.L - 1, -1, -1):
            capi.l2norm_rows_bwd(self.x[k], self.inv[k], self.dS, self.rows, self.ld, T, stream)
            if g is None:
                T, G[0] = G[0], T
                g = G[0]
            else:
                out = G[1] if g is G[0] else G[0]
                capi.spmm_csr(self.planT, g, out, self.ld, d_addend=T, addend_scale=1.0, stream=stream,
                              d_x_row_mask=ds_rows if first else None)
                g, first = out, False
        out = G[1] if g is G[0] else G[0]
        capi.spmm_csr(self.planT, g, out, self.ld, d_addend=self.dS, addend_scale=1.0, d_accum=d_total, stream=stream,
                      d_x_row_mask=ds_rows if first else None)
        return T


class SEPTTrainer:
    """model/ranking/SEPT.py:124-307 on the device.  Variables W = [U; V]; all four views start from W / 2.
    ``train_step_async(..., joint=False)``: recommendation task only (BPR on the preference view + regU l2 of the halved
    tables, Adam #1); ``joint=True``: plus ss_rate x the neighbour-discrimination loss of the three encoders against the
    perturbed-graph view (qrec_sept_ssl_loss_grad), Adam #2 -- two optimizers with their own slots, as in the reference."""

    def __init__(self, U0, V0, adj, friend, sharing, n_layers: int, lr: float, reg: float, ss_rate: float, ins_cnt: int,
                 loss_eps: float = 1e-7, max_unique: int = 4096):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        self.ld = padded_ld(self.d, np.float32)
        self.L, self.lr, self.reg, self.ss_rate, self.k, self.loss_eps = n_layers, lr, reg, ss_rate, ins_cnt, loss_eps
        W0 = np.zeros((self.n, self.ld), np.float32)
        W0[:self.nu, :self.d] = U0; W0[self.nu:, :self.d] = V0
        self.W = DeviceBuffer.from_numpy(W0)
        z = lambda: DeviceBuffer.zeros((self.n, self.ld), np.float32)
        self.E0, self.dE0, self.T, self.G = z(), z(), z(), [z(), z()]
        self.pref, self.aug = _View(self.n, self.ld, n_layers), _View(self.n, self.ld, n_layers)
        self.friend, self.sharing = _View(self.nu, self.ld, n_layers), _View(self.nu, self.ld, n_layers)
        self.pref.set_matrix(adj, split_row=self.nu); self.friend.set_matrix(friend); self.sharing.set_matrix(sharing)
        self.opt = [_Adam(self.W, lr), _Adam(self.W, lr)]          # v1_opt (rec_loss), v2_opt (rec + ss), SEPT.py:267-270
        self.d_loss = DeviceBuffer.zeros(3, np.float64)             # [bpr term, sum W^2, neighbour-discrimination (unscaled)]
        self.row_mask = DeviceBuffer.zeros((self.n + 31) // 32, np.uint32)   # the batch's rows {u, nu+i, nu+j}
        self.max_unique = max_unique
        self.ws = DeviceBuffer(capi.sept_ssl_workspace_bytes(max_unique, self.ld, ins_cnt), np.uint8)
        self.d_labels = None

    def set_perturbed_graph(self, M):
        """the epoch's get_adj_mat(is_subgraph=True) (scipy CSR, N x N)"""
        self.aug.set_matrix(M, split_row=self.nu)

    def _halve(self, stream=None):
        capi.scale_copy(self.E0, self.W, self.n * self.ld, 0.5, stream)

    def train_step_async(self, d_u, d_i, d_j, B: int, joint: bool = False, d_uniq_users=None, n_uu: int = 0, stream=None,
                         keep_labels: bool = False, share=None):
        """``share`` = (offset, count) (one process per GPU, ``self.dp``): BPR on this rank's rows of the step; the
        neighbour-discrimination term, which couples all the step's users, whole on every rank with weight ss_rate / world."""
        dp = getattr(self, "dp", None)
        lo, cnt = (0, B) if share is None else share
        if joint and n_uu > self.max_unique:
            raise ValueError("more unique users in the batch than the SEPT workspace holds")
        if joint and self.aug.plan is None:
            raise RuntimeError("set_perturbed_graph() first")
        self._halve(stream)
        self.row_mask.fill_bytes(0, stream)
        capi.mark_batch_rows(d_u, d_i, d_j, B, self.nu, self.row_mask, stream)   # every S is read at these rows only (the
        mask = self.row_mask                                                      # unique users are among them)
        self.pref.forward(self.E0, stream, last_rows=mask)
        self.pref.dS.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream); self.dE0.fill_bytes(0, stream)
        if cnt:
            off = 4 * lo
            capi.bpr_batch_loss_grad(self.pref.S, 1.0, self.nu, self.n, self.ld, capi.device_ptr(d_u) + off, capi.device_ptr(d_i) + off,
                                     capi.device_ptr(d_j) + off, cnt, self.loss_eps, 0.0, self.pref.dS, self.d_loss, stream, ordered=self.ows)
        capi.sumsq(self.W, capi.F32, self.n, self.d, self.ld, self.d_loss.ptr + 8, stream)     # regU (l2(U/2) + l2(V/2)) = regU sum W^2 / 8
        if joint:
            for v in (self.friend, self.sharing, self.aug):
                v.forward(self.E0, stream, last_rows=mask); v.dS.fill_bytes(0, stream)
            if keep_labels:
                self.d_labels = DeviceBuffer((3, n_uu, self.k), np.int32)
            capi.sept_ssl_loss_grad(self.friend.S, self.sharing.S, self.pref.S, self.aug.S, d_uniq_users, n_uu, self.ld, self.k,
                                    self.ss_rate if dp is None else self.ss_rate / dp.world, self.ws, self.friend.dS, self.sharing.dS, self.pref.dS, self.aug.dS,
                                    self.d_loss.ptr + 16, self.d_labels if keep_labels else None, stream, ordered=self.ows)
            for v in (self.friend, self.sharing, self.aug):
                self.T = v.backward(self.T, self.G, self.dE0, stream, ds_rows=mask)
        self.T = self.pref.backward(self.T, self.G, self.dE0, stream, ds_rows=mask)
        if dp is not None:
            dp.all_reduce(self.dE0); dp.all_reduce(self.d_loss.head_view(1))
        # d/dW = (dE0 + regU E0) / 2 = dE0 / 2 + regU W / 4
        self.opt[1 if joint else 0].step(self.dE0, grad_scale=0.5, stream=stream, grad_l2=self.reg / 4.0)
        self.last_opt = self.opt[1 if joint else 0]

    def gradients(self, before):
        """(dU, dV) of the last step, before Adam; ``before`` = [U; V] when the step started ([n, d])"""
        pad = np.zeros((self.n, self.ld), np.float32); pad[:, :self.d] = before
        g = self.last_opt.applied_gradient(pad)[:, :self.d]
        return g[:self.nu], g[self.nu:]

    def losses(self, stream=None):
        """(rec_loss, ss_rate * neighbor_dis_loss) as the reference prints them (SEPT.py:292, 301)"""
        bpr, ss, nd = self.d_loss.numpy(stream)
        return float(bpr + self.reg * ss / 8.0), float(self.ss_rate * nd)

    def rec_embeddings(self):
        """(rec_user_embeddings, rec_item_embeddings) (SEPT.py:207-208): layer SUM of the preference view"""
        self._halve()
        self.pref.forward(self.E0)
        S = self.pref.S.numpy()[:, :self.d]
        return np.ascontiguousarray(S[:self.nu]), np.ascontiguousarray(S[self.nu:])

    def variables(self):
        W = self.W.numpy()[:, :self.d]
        return W[:self.nu].copy(), W[self.nu:].copy()


# ======================================================================================================================
# MHCN (model/ranking/MHCN.py): motif-induced channel graphs, the multi-channel trainer
# ======================================================================================================================
def mhcn_channel_graphs(n_users: int, n_items: int, uid, iid, ratings, follower, followee):
    """Host side of MHCN.initModel: the three motif-induced user-user adjacencies of buildMotifInducedAdjacencyMatrix
    (MHCN.py:54-85) and the user-item matrix of buildJointAdjacency (:46-52), scipy CSR float32.
    S = follow graph, B = its mutual part, D = S - B its one-way part, Y = purchases.  Social channel: the seven triangle
    motifs over B and D; joint channel: friends (mutual or one-way, symmetrised) who share an item, weighted by the
    number of shared items; purchase channel: non-friends sharing MORE than one item.  Each divided by its row sums."""
    import scipy.sparse as sp
    f32 = np.float32
    fo, fe = np.asarray(follower, np.int64), np.asarray(followee, np.int64)
    uid, iid = np.asarray(uid, np.int64), np.asarray(iid, np.int64)
    S = sp.coo_matrix((np.ones(fo.size, f32), (fo, fe)), shape=(n_users, n_users), dtype=f32)
    Y = sp.coo_matrix((np.ones(uid.size, f32), (uid, iid)), shape=(n_users, n_items), dtype=f32)
    B = S.multiply(S.T)
    D = S - B
    DT = D.T
    sym = lambda C: C + C.T
    m1 = sym((D @ D).multiply(DT))
    m2 = sym((B @ D).multiply(DT) + (D @ B).multiply(DT) + (D @ D).multiply(B))
    m3 = sym((B @ B).multiply(D) + (B @ D).multiply(B) + (D @ B).multiply(B))
    m4 = (B @ B).multiply(B)
    m5 = sym((D @ D).multiply(D) + (D @ DT).multiply(D) + (DT @ D).multiply(D))
    m6 = (D @ B).multiply(D) + (B @ DT).multiply(DT) + (DT @ D).multiply(B)
    m7 = (DT @ B).multiply(DT) + (B @ D).multiply(D) + (D @ DT).multiply(B)
    co = Y @ Y.T
    m8 = co.multiply(B)
    m9 = sym(co.multiply(D))
    m10 = co - m8 - m9
    def by_row_sums(M):
        with np.errstate(divide="ignore"):
            return sp.csr_matrix(M.multiply(1.0 / M.sum(axis=1).reshape(-1, 1)))
    H_s = by_row_sums(m1 + m2 + m3 + m4 + m5 + m6 + m7)        # same association as sum([...]): left to right
    H_j = by_row_sums(m8 + m9)
    H_p = by_row_sums(m10.multiply(m10 > 1))
    # user-item matrix: rating / sqrt(#distinct items of u) / sqrt(#distinct users of i) per training row, python floats
    from math import sqrt
    pairs = np.unique(np.stack([uid, iid], 1), axis=0)
    du = np.bincount(pairs[:, 0], minlength=n_users); di = np.bincount(pairs[:, 1], minlength=n_items)
    vals = np.fromiter((float(r) / sqrt(du[u]) / sqrt(di[i]) for u, i, r in zip(uid.tolist(), iid.tolist(), np.asarray(ratings).tolist())),
                       dtype=np.float64, count=uid.size)
    R = sp.csr_matrix((vals.astype(f32), (uid, iid)), shape=(n_users, n_items))
    return [H_s, H_j, H_p], R


class MHCNTrainer:
    """model/ranking/MHCN.py:93-229 on the device.  ``weights``: dict with the reference's keys (gating1-4, gating_bias1-4,
    sgating1-4, sgating_bias1-4, attention, attention_mat).  One ``train_step_async`` = self-gating of the four channels,
    L layers of {channel attention -> mixed users -> items; three hypergraph convolutions; user-item convolution}, every
    layer l2-normalised into its channel's sum, final attention, BPR on the batch, the hierarchical mutual-information
    loss of the three channels with fresh shuffles, the whole backward pass, Adam on U, V and the 18 weight tensors."""

    W_L2 = 0.001            # reg_loss += 0.001 * l2_loss(weight) for every weight (MHCN.py:211-212)

    def __init__(self, U0, V0, weights, H, R, n_layers: int, lr: float, reg: float, ss_rate: float, loss_eps: float = 1e-7,
                 seed: int = 0):
        self.ows = _ordered_ws()      # parity mode (ordered_reductions()): workspace of the ordered gradient scatters; None = float atomics
        self.nu, self.ni, self.d = U0.shape[0], V0.shape[0], U0.shape[1]
        self.n = self.nu + self.ni
        ld = self.ld = padded_ld(self.d, np.float32)
        self.L, self.lr, self.reg, self.ss_rate, self.loss_eps, self.seed = n_layers, lr, reg, ss_rate, loss_eps, seed
        zu = lambda: DeviceBuffer.zeros((self.nu, ld), np.float32)
        zi = lambda: DeviceBuffer.zeros((self.ni, ld), np.float32)
        pad2 = lambda w: np.pad(np.asarray(w, np.float32), ((0, ld - self.d), (0, ld - self.d)))
        pad1 = lambda b: np.pad(np.asarray(b, np.float32).reshape(-1), (0, ld - self.d))
        self.U = DeviceBuffer.from_numpy(np.pad(np.asarray(U0, np.float32), ((0, 0), (0, ld - self.d))))
        self.V = DeviceBuffer.from_numpy(np.pad(np.asarray(V0, np.float32), ((0, 0), (0, ld - self.d))))
        self.w, self.g, self.opt = {}, {}, {}
        for key, val in weights.items():
            arr = pad1(val) if np.asarray(val).shape[0] == 1 else pad2(val)
            self.w[key] = DeviceBuffer.from_numpy(arr)
            self.g[key] = DeviceBuffer.zeros(arr.shape, np.float32)
            self.opt[key] = _Adam(self.w[key], lr)
        self.optU, self.optV = _Adam(self.U, lr), _Adam(self.V, lr)
        self.planH = [SpmmPlan(*_csr_triple(h), ld) for h in H]
        self.planHT = [SpmmPlan(*_csr_triple(h.T), ld) for h in H]
        self.planR, self.planRT = SpmmPlan(*_csr_triple(R), ld), SpmmPlan(*_csr_triple(R.T), ld)
        L = n_layers
        # forward state
        self.G, self.Gs = [zu() for _ in range(4)], [zu() for _ in range(4)]          # gated tables, their sigmoids
        self.c = [[self.G[k] for k in range(3)]] + [[zu() for _ in range(3)] for _ in range(L)]
        self.s = [self.G[3]] + [zu() for _ in range(L)]
        self.t = [self.V] + [zi() for _ in range(L)]
        self.inv_c = [[DeviceBuffer.zeros(self.nu, np.float32) for _ in range(3)] for _ in range(L)]
        self.inv_s = [DeviceBuffer.zeros(self.nu, np.float32) for _ in range(L)]
        self.inv_t = [DeviceBuffer.zeros(self.ni, np.float32) for _ in range(L)]
        self.sum_c, self.sum_s = [zu() for _ in range(3)], zu()
        self.F = DeviceBuffer.zeros((self.n, ld), np.float32)                       # [final users ; final items (= sum_t)]
        self.dF = DeviceBuffer.zeros((self.n, ld), np.float32)
        self.score = [DeviceBuffer.zeros((self.nu, 4), np.float32) for _ in range(L + 1)]
        self.v = DeviceBuffer.zeros(256, np.float32); self.dv = DeviceBuffer.zeros(capi.channel_attention_scratch_floats(), np.float32)
        self.mixed = zu()
        # self-supervision state
        self.SG, self.SGs, self.edge, self.dem, self.dedge, self.dSG, self.Q = zu(), zu(), zu(), zu(), zu(), zu(), zu()
        self.hss_ws = DeviceBuffer(capi.hss_scratch_bytes(self.nu), np.uint8)
        self.wg_ws = DeviceBuffer(capi.buir_wgrad_scratch_bytes(ld), np.uint8)
        self.perm_ws = DeviceBuffer(capi.random_permutations_scratch_bytes(self.nu, 9), np.uint8)
        self.rowp, self.rowq = DeviceBuffer((9, self.nu), np.int32), DeviceBuffer((9, self.nu), np.int32)   # (p1, p2, p3) x 3 channels; inverses
        self.colp, self.colq = DeviceBuffer((6, self.d), np.int32), DeviceBuffer((6, self.d), np.int32)    # (k2, k3) x 3 channels; inverses
        # backward state
        self.dsum_c, self.dsum_s = [zu() for _ in range(3)], zu()
        self.gc = [[zu() for _ in range(3)] for _ in range(2)]
        self.gs, self.gt = [zu(), zu()], [zi(), zi()]
        self.Tu, self.Ti, self.dmixed, self.dU = zu(), zi(), zu(), zu()
        self.d_loss = DeviceBuffer.zeros(2, np.float64)             # [rec, self-supervised (unscaled)]
        self.step_no = 0

    # ---- pointers into the stacked final table ---------------------------------------------------------------
    @property
    def _FI(self):
        return self.F.ptr + self.nu * self.ld * 4

    @property
    def _dFI(self):
        return self.dF.ptr + self.nu * self.ld * 4

    def forward(self, stream=None):
        """fills F = [final_user_embeddings ; final_item_embeddings] (MHCN.py:130-174)"""
        nu, ni, ld, L, w = self.nu, self.ni, self.ld, self.L, self.w
        for k in range(4):
            capi.gate_fwd(self.U, w[f"gating{k + 1}"], w[f"gating_bias{k + 1}"], nu, ld, self.G[k], self.Gs[k], stream)
        for k in range(3):
            self.sum_c[k].copy_from(self.G[k], stream)
        self.sum_s.copy_from(self.G[3], stream)
        capi.memcpy_d2d(self._FI, self.V.ptr, ni * ld * 4, stream)
        for l in range(1, L + 1):
            capi.channel_attention_fwd(self.c[l - 1], w["attention"], w["attention_mat"], self.s[l - 1], nu, ld, self.v,
                                       self.score[l - 1], self.mixed, stream)
            for k in range(3):
                capi.spmm_csr(self.planH[k], self.c[l - 1][k], self.c[l][k], ld, stream=stream)
                capi.l2norm_rows_accum(self.c[l][k], nu, ld, self.sum_c[k], self.inv_c[l - 1][k], stream)
            capi.spmm_csr(self.planRT, self.mixed, self.t[l], ld, stream=stream)
            capi.l2norm_rows_accum(self.t[l], ni, ld, self._FI, self.inv_t[l - 1], stream)
            capi.spmm_csr(self.planR, self.t[l - 1], self.s[l], ld, stream=stream)
            capi.l2norm_rows_accum(self.s[l], nu, ld, self.sum_s, self.inv_s[l - 1], stream)
        capi.channel_attention_fwd(self.sum_c, w["attention"], w["attention_mat"], self.sum_s, nu, ld, self.v, self.score[L],
                                   self.F, stream)

    def _draw_shuffles(self, perms, stream):
        """device pointers (p1, p1inv, p2, p2inv, k2, k2inv, p3, p3inv, k3, k3inv) per channel.  ``perms`` (tests): host
        arrays [(p1, k2, p2, k3, p3)] x 3; otherwise fresh uniform shuffles are drawn on the device."""
        if perms is not None:
            inv = lambda p: np.argsort(p).astype(np.int32)
            rows = [np.asarray(perms[k][j], np.int32) for k in range(3) for j in (0, 2, 4)]
            cols = [np.asarray(perms[k][j], np.int32) for k in range(3) for j in (1, 3)]
            self.rowp.upload(np.stack(rows), stream); self.rowq.upload(np.stack([inv(p) for p in rows]), stream)
            self.colp.upload(np.stack(cols), stream); self.colq.upload(np.stack([inv(p) for p in cols]), stream)
        else:       # all nine row shuffles of the step from one sort, the six column shuffles from one small kernel
            capi.random_permutations(self.nu, 9, self.seed, 2 * self.step_no, self.perm_ws, self.rowp, self.rowq, stream)
            capi.small_permutations(self.d, 6, self.seed, 2 * self.step_no + 1, self.colp, self.colq, stream)
        rb, cb = 4 * self.nu, 4 * self.d
        out = []
        for k in range(3):
            rp = [self.rowp.ptr + (3 * k + j) * rb for j in range(3)]; rq = [self.rowq.ptr + (3 * k + j) * rb for j in range(3)]
            cp = [self.colp.ptr + (2 * k + j) * cb for j in range(2)]; cq = [self.colq.ptr + (2 * k + j) * cb for j in range(2)]
            out.append((rp[0], rq[0], rp[1], rq[1], cp[0], cq[0], rp[2], rq[2], cp[1], cq[1]))
        return out

    def train_step_async(self, d_u, d_i, d_j, B: int, perms=None, stream=None):
        """With ``self.dp`` (one process per GPU) d_u/d_i/d_j are this rank's share of the step's rows; the
        self-supervised term does not depend on the batch, so every rank forms it whole with weight ss_rate / world."""
        nu, ni, ld, d, L, w, g = self.nu, self.ni, self.ld, self.d, self.L, self.w, self.g
        dp = getattr(self, "dp", None)
        ss_rate = self.ss_rate if dp is None else self.ss_rate / dp.world
        self.forward(stream)
        self.dF.fill_bytes(0, stream); self.d_loss.fill_bytes(0, stream)
        g["attention"].fill_bytes(0, stream); g["attention_mat"].fill_bytes(0, stream)
        for k in (1, 2, 3, 4):        # sgating4 is created but unused: its gradient is its L2 term alone
            g[f"sgating{k}"].fill_bytes(0, stream); g[f"sgating_bias{k}"].fill_bytes(0, stream)
        if B:
            capi.bpr_batch_loss_grad(self.F, 1.0, nu, self.n, ld, d_u, d_i, d_j, B, self.loss_eps, 0.0, self.dF, self.d_loss, stream, ordered=self.ows)
        # hierarchical self-supervision of the three channels (MHCN.py:176-178, 184-206)
        for k, shuffles in enumerate(self._draw_shuffles(perms, stream)):
            Ws, bs = w[f"sgating{k + 1}"], w[f"sgating_bias{k + 1}"]
            capi.gate_fwd(self.F, Ws, bs, nu, ld, self.SG, self.SGs, stream)
            capi.spmm_csr(self.planH[k], self.SG, self.edge, ld, stream=stream)
            capi.hss_loss_grad(self.SG, self.edge, nu, d, ld, shuffles, ss_rate, self.hss_ws, self.dem, self.dedge,
                               self.d_loss.ptr + 8, stream)
            capi.spmm_csr(self.planHT[k], self.dedge, self.dSG, ld, d_addend=self.dem, addend_scale=1.0, stream=stream)
            capi.gate_bwd(self.F, self.SGs, self.dSG, Ws, nu, d, ld, self.Q, self.dF, True, stream=stream)
            capi.buir_wgrad(self.F, self.Q, nu, ld, self.wg_ws, g[f"sgating{k + 1}"], g[f"sgating_bias{k + 1}"], stream)
        # final aggregation (MHCN.py:173-174): dsum_c, dsum_s; dsum_t is the item half of dF
        capi.channel_attention_bwd(self.dF, self.sum_c, self.score[L], self.v, w["attention"], w["attention_mat"], nu, ld,
                                   self.dsum_c, False, self.dsum_s, False, self.dv, g["attention"], g["attention_mat"], stream)
        a = 0
        for k in range(3):
            capi.l2norm_rows_bwd(self.c[L][k], self.inv_c[L - 1][k], self.dsum_c[k], nu, ld, self.gc[a][k], stream)
        capi.l2norm_rows_bwd(self.s[L], self.inv_s[L - 1], self.dsum_s, nu, ld, self.gs[a], stream)
        capi.l2norm_rows_bwd(self.t[L], self.inv_t[L - 1], self._dFI, ni, ld, self.gt[a], stream)
        for l in range(L, 0, -1):          # gradients of level l -> level l-1
            b = 1 - a
            first = l - 1 == 0             # level 0 enters the sums un-normalised
            capi.spmm_csr(self.planR, self.gt[a], self.dmixed, ld, stream=stream)                      # t^(l) = R^T mixed^(l)
            if first:
                base_t = self._dFI
            else:
                capi.l2norm_rows_bwd(self.t[l - 1], self.inv_t[l - 2], self._dFI, ni, ld, self.Ti, stream); base_t = self.Ti
            capi.spmm_csr(self.planRT, self.gs[a], self.gt[b], ld, d_addend=base_t, addend_scale=1.0, stream=stream)   # s^(l) = R t^(l-1)
            for k in range(3):                                                                          # c^(l) = H c^(l-1)
                if first:
                    base_c = self.dsum_c[k]
                else:
                    capi.l2norm_rows_bwd(self.c[l - 1][k], self.inv_c[l - 2][k], self.dsum_c[k], nu, ld, self.Tu, stream); base_c = self.Tu
                capi.spmm_csr(self.planHT[k], self.gc[a][k], self.gc[b][k], ld, d_addend=base_c, addend_scale=1.0, stream=stream)
            if first:
                self.gs[b].copy_from(self.dsum_s, stream)
            else:
                capi.l2norm_rows_bwd(self.s[l - 1], self.inv_s[l - 2], self.dsum_s, nu, ld, self.gs[b], stream)
            # mixed^(l) = attention(c^(l-1)) + s^(l-1) / 2
            capi.channel_attention_bwd(self.dmixed, self.c[l - 1], self.score[l - 1], self.v, w["attention"], w["attention_mat"], nu, ld,
                                       self.gc[b], True, self.gs[b], True, self.dv, g["attention"], g["attention_mat"], stream)
            a = b
        # self-gating of the four channels (MHCN.py:131-134)
        for k in range(4):
            dG = self.gc[a][k] if k < 3 else self.gs[a]
            capi.gate_bwd(self.U, self.Gs[k], dG, w[f"gating{k + 1}"], nu, d, ld, self.Q, self.dU, k > 0, stream=stream)
            capi.buir_wgrad(self.U, self.Q, nu, ld, self.wg_ws, g[f"gating{k + 1}"], g[f"gating_bias{k + 1}"], stream)
        if dp is not None:      # table and weight gradients: sums over the ranks' shares (+ the 1/world self-supervised parts)
            for buf in [self.dU, self.gt[a], self.d_loss.head_view(1)] + list(g.values()):
                dp.all_reduce(buf)
        self.optU.step(self.dU, stream=stream, grad_l2=self.reg)
        self.optV.step(self.gt[a], stream=stream, grad_l2=self.reg)
        for key, opt in self.opt.items():
            opt.step(g[key], stream=stream, grad_l2=self.W_L2)
        self.step_no += 1

    def gradients(self, before):
        """the last step's gradients by the reference's variable keys (+ "U", "V"), before Adam; ``before``: the same keys ->
        the variables' values when the step started (every variable has an L2 term: MHCN.py:209-212)"""
        d, ld = self.d, self.ld

        def padded(a, like):
            out = np.zeros(like.shape, np.float32); a = np.asarray(a, np.float32)
            if len(like.shape) == 1:
                out[:a.size] = a.reshape(-1)
            else:
                out[:a.shape[0], :a.shape[1]] = a
            return out
        out = {}
        for key, opt in list(self.opt.items()) + [("U", self.optU), ("V", self.optV)]:
            g = opt.applied_gradient(padded(before[key], opt.theta))
            out[key] = g[None, :d].copy() if g.ndim == 1 else (g[:, :d].copy() if key in ("U", "V") else g[:d, :d].copy())
        return out

    def losses(self, stream=None):
        """(rec_loss -- what the reference prints, MHCN.py:223-225 --, ss_loss unscaled)"""
        rec, ss = self.d_loss.numpy(stream)
        return float(rec), float(ss)

    def final_embeddings(self):
        """(final_user_embeddings, final_item_embeddings) (MHCN.py:172-174, 226)"""
        self.forward()
        F = self.F.numpy()[:, :self.d]
        return np.ascontiguousarray(F[:self.nu]), np.ascontiguousarray(F[self.nu:])

    def parameters(self):
        d = self.d
        out = {k: (b.numpy()[:d, :d].copy() if len(b.shape) == 2 else b.numpy()[None, :d].copy()) for k, b in self.w.items()}
        out["U"], out["V"] = self.U.numpy()[:, :d].copy(), self.V.numpy()[:, :d].copy()
        return out
