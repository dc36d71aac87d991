"""Batched full-rank evaluation on the device (base/recommender.py:143-150,
util/qmath.py:134-146): scores by MFMA, rated items masked to 0, the reference's heap
top-N emulated lane-per-user.  ``DeviceRanker.topk`` returns ids/scores for many users at
once; nothing is computed on the host."""
from __future__ import annotations

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
        self.dU = DeviceBuffer.from_numpy(np.ascontiguousarray(U))
        self.dV = DeviceBuffer.from_numpy(np.ascontiguousarray(V))
        self._scratch = self._d_ids = self._d_sc = None
        self._cap = (0, 0)
        self.rated = None
        if rated is not None:
            self.rated = (DeviceBuffer.from_numpy(rated.indptr.astype(np.int64)),
                          DeviceBuffer.from_numpy(rated.indices.astype(np.int32)))

    def update_tables(self, U: np.ndarray, V: np.ndarray):
        """New embeddings of the same shape/dtype (per-epoch evaluation): re-upload only."""
        if U.shape != (self.n_users, self.d) or V.shape != (self.n_items, self.d) or U.dtype != self.dtype or V.dtype != self.dtype:
            raise ValueError("update_tables: shape/dtype differ from the ranker's")
        self.dU.upload(np.ascontiguousarray(U)); self.dV.upload(np.ascontiguousarray(V))

    def topk(self, user_ids: np.ndarray, N: int):
        """(ids int32 [n, N], scores [n, N]) in the reference's order (descending score,
        heap order among ties)."""
        user_ids = np.ascontiguousarray(user_ids, dtype=np.int32)
        n = user_ids.size
        ids = np.empty((n, N), dtype=np.int32)
        scores = np.empty((n, N), dtype=self.dtype)
        if n == 0:
            return ids, scores
        if user_ids.min() < 0 or user_ids.max() >= self.n_users:
            raise ValueError("user id out of range")
        per_user = capi.score_topk_scratch_bytes(self.code, self.n_items, 64) // 64
        batch = int(max(64, min(n, (_SCRATCH_BUDGET // max(per_user, 1)) // 64 * 64)))
        nb = min(batch, n)
        if self._cap[0] < nb or self._cap[1] < N:      # the 4-10 GB score block is kept across calls
            self._scratch = DeviceBuffer(capi.score_topk_scratch_bytes(self.code, self.n_items, nb), np.uint8)
            self._d_ids = DeviceBuffer((nb, N), np.int32); self._d_sc = DeviceBuffer((nb, N), self.dtype)
            self._cap = (nb, N)
        scratch, d_ids, d_sc = self._scratch, self._d_ids, self._d_sc
        if d_ids.shape[1] != N:                         # smaller N than the cached buffers: use exact-size views
            d_ids = DeviceBuffer((nb, N), np.int32); d_sc = DeviceBuffer((nb, N), self.dtype)
        for s in range(0, n, batch):
            chunk = user_ids[s:s + batch]
            d_users = DeviceBuffer.from_numpy(chunk)
            capi.score_topk(self.dU, self.dV, self.code, self.d, self.d, self.n_items, d_users, chunk.size,
                            self.rated[0] if self.rated else None, self.rated[1] if self.rated else None,
                            N, scratch, d_ids, d_sc)
            ids[s:s + chunk.size] = d_ids.numpy()[:chunk.size]
            scores[s:s + chunk.size] = d_sc.numpy()[:chunk.size]
        return ids, scores
