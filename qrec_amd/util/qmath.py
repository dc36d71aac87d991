"""Host math helpers of the hot path (util/qmath.py:127-146)."""
from __future__ import annotations

import heapq
from math import exp


def sigmoid(val: float) -> float:
    return 1 / (1 + exp(-val))


def find_k_largest(K: int, candidates):
    """Top-K by the reference's procedure: a min-heap of ``(score, id)`` seeded with the
    first K entries, strict ``>`` replacement, then a stable descending sort by score
    (util/qmath.py:134-146).  Kept on the host for odd cases (ties at the cut, cold users);
    the batched device path is qrec_amd.ranking."""
    heap = [(score, iid) for iid, score in enumerate(candidates[:K])]
    heapq.heapify(heap)
    for iid in range(K, len(candidates)):
        score = candidates[iid]
        if score > heap[0][0]:
            heapq.heapreplace(heap, (score, iid))
    heap.sort(key=lambda pair: pair[0], reverse=True)
    return [iid for _, iid in heap], [score for score, _ in heap]
