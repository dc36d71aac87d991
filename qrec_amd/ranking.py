"""Batched full-rank evaluation on the device (base/recommender.py:143-150,
util/qmath.py:134-146): scores by MFMA, rated items masked to 0, the reference's heap
top-N emulated lane-per-user.  ``DeviceRanker.topk`` returns ids/scores for many users at
once; nothing is computed on the host.  With ``set_test`` + ``cuts`` it also returns each user's hit count and
DCG sum (util/measure.py:15-21,70-82) from the lists while they are still on the device, so that the per-epoch
evaluation of the graph models never builds 31 k Python lists (0.8 s of host time per evaluation at the
Yelp2018 shape vs 10 ms of kernels)."""
from __future__ import annotations

import math

import numpy as np

from . import capi
from .capi import DeviceBuffer
from .interactions import CSR

_SCRATCH_BUDGET = 8 << 30   # bytes of transposed score block per batch (HBM is 288 GB)


class DeviceRanker:
    def __init__(self, U: np.ndarray, V: np.ndarray, rated: CSR | None = None):
        """U [users, d], V [items, d] host arrays of the same float dtype (fp64 for the
        numpy-path models, fp32 for the TF-path ones); ``rated`` = train items per user."""
        if U.dtype != V.dtype or U.dtype not in (np.float32, np.float64):
            raise TypeError("U and V must both be float32 or both float64")
        self.dtype = U.dtype
        self.code = capi.F64 if self.dtype == np.float64 else capi.F32
        self.d = U.shape[1]
        assert V.shape[1] == self.d
        self.n_users, self.n_items = U.shape[0], V.shape[0]
        slot = 16 if self.dtype == np.float64 else 32          # MFMA k-slot width of the scoring kernels
        self.ld = -(-self.d // slot) * slot                     # zero-padded row stride (include/qrec_hip.h)
        self.dU = DeviceBuffer.from_numpy(self._pad(U))
        self.dV = DeviceBuffer.from_numpy(self._pad(V))
        self._scratch = self._d_ids = self._d_sc = None
        self._cap = (0, 0)
        self.rated = None
        self.test = None
        if rated is not None:
            rated = rated.sorted_rows()        # the fused route looks rated items up by bisection (include/qrec_hip.h)
            self.rated = (DeviceBuffer.from_numpy(rated.indptr.astype(np.int64)),
                          DeviceBuffer.from_numpy(rated.indices.astype(np.int32) if rated.indices.size else np.zeros(1, np.int32)))

    def _pad(self, a: np.ndarray) -> np.ndarray:
        if a.shape[1] == self.ld:
            return np.ascontiguousarray(a)
        out = np.zeros((a.shape[0], self.ld), dtype=self.dtype)
        out[:, :self.d] = a
        return out

    def update_tables(self, U: np.ndarray, V: np.ndarray):
        """New embeddings of the same shape/dtype (per-epoch evaluation): re-upload only."""
        if U.shape != (self.n_users, self.d) or V.shape != (self.n_items, self.d) or U.dtype != self.dtype or V.dtype != self.dtype:
            raise ValueError("update_tables: shape/dtype differ from the ranker's")
        self.dU.upload(self._pad(U)); self.dV.upload(self._pad(V))

    def set_test(self, test: CSR):
        """held-out items per user (CSR over ALL users of the table, ids ascending inside a row)"""
        srt = test.sorted_rows()
        if srt.indptr.size != self.n_users + 1:
            raise ValueError("set_test: the CSR must have one row per user of the table")
        self.test = (DeviceBuffer.from_numpy(srt.indptr.astype(np.int64)),
                     DeviceBuffer.from_numpy(srt.indices.astype(np.int32) if srt.indices.size else np.zeros(1, np.int32)))

    def topk(self, user_ids: np.ndarray, N: int, cuts=None, want_lists: bool = True):
        """(ids int32 [n, N], scores [n, N]) in the reference's order (descending score,
        heap order among ties).  With ``cuts`` (list of list lengths <= N; needs ``set_test``) a third
        value {cut: (hits int32 [n], dcg float64 [n])} comes back, computed from the device-resident lists;
        ``want_lists=False`` then skips the read-back of ids and scores."""
        user_ids = np.ascontiguousarray(user_ids, dtype=np.int32)
        n = user_ids.size
        ids = np.empty((n, N), dtype=np.int32) if want_lists else None
        scores = np.empty((n, N), dtype=self.dtype) if want_lists else None
        per_cut = None
        if cuts is not None:
            if self.test is None:
                raise RuntimeError("topk(cuts=...) needs set_test() first")
            if any(c < 1 or c > N for c in cuts):
                raise ValueError("cuts must lie in 1..N")
            per_cut = {c: (np.zeros(n, np.int32), np.zeros(n, np.float64)) for c in cuts}
            d_disc = DeviceBuffer.from_numpy(np.array([1.0 / math.log(pos + 2) for pos in range(N)], np.float64))
        if n == 0:
            return (ids, scores) if cuts is None else (ids, scores, per_cut)
        if user_ids.min() < 0 or user_ids.max() >= self.n_users:
            raise ValueError("user id out of range")
        # users per call: as many as the scratch budget allows (fused route: ~15 KB per user at the Yelp2018 shape,
        # block route: the whole score column, 150 KB)
        per_user = capi.score_topk_scratch_bytes(self.code, self.n_items, 4096, self.ld, N) // 4096
        batch = int(max(64, min(n, (_SCRATCH_BUDGET // max(per_user, 1)) // 64 * 64)))
        nb = min(batch, n)
        # kept across calls (per-epoch evaluation).  The byte count depends on the evaluation route the library picks from
        # QREC_EVAL_* at every call (bf16 copies, list capacity): it is asked again each time and the scratch re-made when the
        # cached one is too small for the route now in force
        need = capi.score_topk_scratch_bytes(self.code, self.n_items, nb, self.ld, N)
        if self._cap[0] < nb or self._cap[1] != N or self._scratch.nbytes < need:
            self._scratch = DeviceBuffer(need, np.uint8)
            self._d_ids = DeviceBuffer((nb, N), np.int32); self._d_sc = DeviceBuffer((nb, N), self.dtype)
            self._cap = (nb, N)
        scratch, d_ids, d_sc = self._scratch, self._d_ids, self._d_sc
        if d_ids.shape[1] != N:                         # smaller N than the cached buffers: use exact-size views
            d_ids = DeviceBuffer((nb, N), np.int32); d_sc = DeviceBuffer((nb, N), self.dtype)
        for s in range(0, n, batch):
            chunk = user_ids[s:s + batch]
            d_users = DeviceBuffer.from_numpy(chunk)
            capi.score_topk(self.dU, self.dV, self.code, self.d, self.ld, self.n_items, d_users, chunk.size,
                            self.rated[0] if self.rated else None, self.rated[1] if self.rated else None,
                            N, scratch, d_ids, d_sc)
            if want_lists:
                ids[s:s + chunk.size] = d_ids.numpy()[:chunk.size]
                scores[s:s + chunk.size] = d_sc.numpy()[:chunk.size]
            if per_cut:
                d_hits, d_dcg = DeviceBuffer(chunk.size, np.int32), DeviceBuffer(chunk.size, np.float64)
                for c, (hits, dcg) in per_cut.items():
                    capi.rank_hits(d_ids, chunk.size, N, c, d_users, self.test[0], self.test[1], d_disc, d_hits, d_dcg)
                    hits[s:s + chunk.size] = d_hits.numpy(); dcg[s:s + chunk.size] = d_dcg.numpy()
        return (ids, scores) if cuts is None else (ids, scores, per_cut)


def ranking_measure_strings(test_lens, per_n: dict, Ns) -> list:
    """The strings of Measure.rankingMeasure (util/measure.py:24-49) from per-user hit counts and DCG sums:
    ``per_n[n] = (hits, dcg)`` sequences in testSet_u order, ``test_lens[k] = len(testSet_u[user_k])``.  Same
    operations in the same order as the reference: its ``sum()`` calls are left-to-right fp64 additions, which is what
    ``np.cumsum(...)[-1]`` computes (a scan, not numpy's pairwise ``sum``), so the digits are the same."""
    # (no interpreter-version guard here: the emulation is numpy's, its digits do not depend on what the running CPython's sum() does)
    out = []
    lens = np.asarray(test_lens, dtype=np.int64)
    n_users = lens.size
    for n in Ns:
        hits = np.asarray(per_n[n][0], dtype=np.int64); dcg = np.asarray(per_n[n][1], dtype=np.float64)
        prec = int(hits.sum()) / (n_users * n)                                   # exact integer sum
        rec = float(np.cumsum(hits / lens)[-1]) / n_users if n_users else 0.0
        f1 = 2 * prec * rec / (prec + rec) if (prec + rec) != 0 else 0
        prefix, acc = [0], 0
        for pos in range(n):
            acc += 1.0 / math.log(pos + 2)
            prefix.append(acc)
        idcg = np.asarray(prefix, dtype=np.float64)[np.minimum(lens, n)]
        ndcg = float(np.cumsum(dcg / idcg)[-1]) / n_users if n_users else 0.0
        out += ["Top " + str(n) + "\n", "Precision:" + str(prec) + "\n", "Recall:" + str(rec) + "\n",
                "F1:" + str(f1) + "\n", "NDCG:" + str(ndcg) + "\n"]
    return out
