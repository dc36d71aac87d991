"""Array form of the reference data model (data/rating.py) -- what the kernels consume.

The reference keeps ``trainSet_u[user][item] = rating`` dict-of-dicts
(data/rating.py:33-67).  Iteration order of those dicts *is* part of the hot path's
contract: BPR walks users in id order and each user's items in dict insertion order
(model/ranking/BPR.py:31-34), duplicates of a (user,item) row keep their first position
and their last rating.  ``Interactions`` reproduces exactly that order as CSR arrays.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def first_appearance_ids(names: np.ndarray):
    """ids in first-appearance order (data/rating.py:48-54).  Returns (ids, uniq_names)
    where uniq_names[k] is the name with id k."""
    uniq, first, inv = np.unique(names, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(order.size, dtype=np.int64)
    rank[order] = np.arange(order.size)
    return rank[inv].astype(np.int32), uniq[order]


@dataclass
class CSR:
    indptr: np.ndarray   # int64 [rows+1]
    indices: np.ndarray  # int32 [nnz]
    values: np.ndarray | None = None

    @property
    def n_rows(self) -> int:
        return self.indptr.size - 1

    @property
    def nnz(self) -> int:
        return int(self.indices.size)

    def row_ids(self) -> np.ndarray:
        return np.repeat(np.arange(self.n_rows, dtype=np.int32), np.diff(self.indptr))

    def sorted_rows(self) -> "CSR":
        """Same rows with column ids ascending inside each row (membership search)."""
        r = self.row_ids().astype(np.int64)
        order = np.lexsort((self.indices, r))
        return CSR(self.indptr, self.indices[order],
                   None if self.values is None else self.values[order])


def dedup_user_item(uid: np.ndarray, iid: np.ndarray, rating: np.ndarray, n_items: int):
    """dict semantics of ``trainSet_u[u][i] = r`` over rows in file order: position of
    the FIRST occurrence, rating of the LAST.  Returns (u, i, r) in global first-occurrence
    order (which, grouped stably by u, is each user's dict order)."""
    key = uid.astype(np.int64) * np.int64(n_items) + iid.astype(np.int64)
    if key.size:                                      # the usual file has no duplicate (user, item) rows: a plain sort
        ks = np.sort(key)                             # (vectorised, unstable is fine) settles that several times faster
        if not (ks[1:] == ks[:-1]).any():             # than the stable argsort the general case needs
            return uid, iid, rating
    order = np.argsort(key, kind="stable")            # one sort; equal keys stay in file order
    ks = key[order]
    start = np.ones(ks.size, dtype=bool)
    start[1:] = ks[1:] != ks[:-1]
    if start.all():                                   # no duplicate (user, item) rows: nothing to do
        return uid, iid, rating
    first = order[start]                              # first occurrence of each key
    last = order[np.append(start[1:], True)]          # last occurrence
    by_first = np.argsort(first, kind="stable")
    first, last = first[by_first], last[by_first]
    return uid[first], iid[first], rating[last]


def user_item_csr(uid, iid, rating, n_users: int, n_items: int, min_rating: float | None = None,
                  assume_unique: bool = False) -> CSR:
    """CSR of ``trainSet_u`` (``min_rating=None``) or of BPR's ``PositiveSet``
    (``min_rating=1``, model/ranking/BPR.py:21-25) in the reference's iteration order."""
    u, i, r = np.asarray(uid), np.asarray(iid), np.asarray(rating, dtype=np.float64)
    if not assume_unique:
        u, i, r = dedup_user_item(u, i, r, n_items)
    if min_rating is not None:
        keep = r >= min_rating
        u, i, r = u[keep], i[keep], r[keep]
    order = np.argsort(u, kind="stable")
    counts = np.bincount(u, minlength=n_users)
    indptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return CSR(indptr, np.ascontiguousarray(i[order], dtype=np.int32),
               np.ascontiguousarray(r[order], dtype=np.float64))
