"""``RatingRows``: the reference's ``trainingData`` / ``testData`` list of ``[userName, itemName, rating]`` rows
(util/io.py:69-72) held as three arrays plus the two name tables.  It reads like the list it replaces (len,
iteration, indexing, slicing, equality with a list) and materialises Python rows only when somebody asks for
them; the native loader (``qrec_ratings_load``), the ``-ap`` / ``-cv`` splits and ``Rating`` work on the arrays."""
from __future__ import annotations

from collections.abc import Sequence

import numpy as np


class RatingRows(Sequence):
    __slots__ = ("user_idx", "item_idx", "rating", "user_names", "item_names")

    def __init__(self, user_idx, item_idx, rating, user_names, item_names):
        self.user_idx = np.ascontiguousarray(user_idx, dtype=np.int32)
        self.item_idx = np.ascontiguousarray(item_idx, dtype=np.int32)
        self.rating = np.ascontiguousarray(rating, dtype=np.float64)
        self.user_names, self.item_names = user_names, item_names     # file-level first-appearance tables (shared)

    def __len__(self):
        return int(self.user_idx.size)

    def take(self, index) -> "RatingRows":
        """rows selected by an index array, a boolean mask or a slice (name tables are shared, ids unchanged)"""
        return RatingRows(self.user_idx[index], self.item_idx[index], self.rating[index], self.user_names, self.item_names)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return self.take(k)
        k = int(k)
        return [self.user_names[self.user_idx[k]], self.item_names[self.item_idx[k]], float(self.rating[k])]

    def __iter__(self):
        un, inn = self.user_names, self.item_names
        for u, i, r in zip(self.user_idx.tolist(), self.item_idx.tolist(), self.rating.tolist()):
            yield [un[u], inn[i], r]

    def to_list(self) -> list:
        return list(iter(self))

    def __eq__(self, other):
        if isinstance(other, RatingRows):
            other = other.to_list()
        return self.to_list() == other

    __hash__ = None

    def __repr__(self):
        return f"RatingRows({len(self)} rows, {len(self.user_names)} users, {len(self.item_names)} items in the file)"

    # ---- id assignment of a subset, as Rating._ingest does it (data/rating.py:48-54) ------------------
    @staticmethod
    def first_appearance(idx: np.ndarray, n_names: int):
        """(order, remap): ``order[k]`` = file-level id of the k-th distinct value in order of first appearance in
        ``idx``; ``remap[file id]`` = k (or -1 when the value does not occur)."""
        remap = np.full(n_names, -1, dtype=np.int32)
        if idx.size == 0:
            return np.zeros(0, np.int64), remap
        # first[v] = position of v's first occurrence: assign positions back to front -- of repeated indices the last
        # assignment stands, which is the earliest position (one pass instead of np.unique's sort of all rows)
        first = np.full(n_names, idx.size, dtype=np.int64)
        pos = np.arange(idx.size, dtype=np.int64)
        first[idx[::-1]] = pos[::-1]
        if (first[idx] > pos).any():                  # numpy does not promise the order of repeated assignments: checked, not assumed
            uniq, f = np.unique(idx, return_index=True)
            first[:] = idx.size
            first[uniq] = f
        present = np.flatnonzero(first < idx.size)
        order = present[np.argsort(first[present], kind="stable")].astype(idx.dtype, copy=False)
        remap[order] = np.arange(order.size, dtype=np.int32)
        return order, remap
