"""The identity the evaluation's exact walk uses on the GPU (KeyHeap::sift in qrec_amd/csrc/eval_topk.hip): CPython's
heapq._siftup(heap, pos) -- bubble the smaller child up to a leaf, then sift the item back down -- nets out to an insertion
along the smaller-child path: the path's entries grow downwards, the item passes exactly those smaller than itself (each moves
up one level) and lands where the next one is larger.  Modelled here in Python next to heapq itself: heapify and heapreplace
through the path form leave the array in the same state as heapq's, element for element, ties in the scores included (tuples
(score, id) are distinct, so the order is total).  The device version of the same check: tools/ubench/exact_walk_probe.hip."""
import heapq
import random

import pytest


def sift_path_form(h, pos, item):
    """heap[pos] is taken to hold ``item``; the subtree below pos is a heap"""
    n = len(h)
    cur = pos
    while True:
        c1, c2 = 2 * cur + 1, 2 * cur + 2
        if c1 >= n:
            break
        nx = c2 if (c2 < n and not h[c1] < h[c2]) else c1            # heapq: the right child unless left < right
        if not h[nx] < item:
            break
        h[cur] = h[nx]
        cur = nx
    h[cur] = item


@pytest.mark.parametrize("k", [1, 2, 3, 7, 20, 21, 63, 64])
def test_path_form_equals_heapq_for_heapify_and_heapreplace(k):
    rng = random.Random(k)
    for trial in range(30):
        levels = rng.choice([3, 10, 1000])                            # few distinct scores: ties everywhere
        items = [(rng.randrange(levels) / levels, i) for i in range(k + 400)]
        a = items[:k]; b = list(a)
        heapq.heapify(a)
        for t in range(k // 2 - 1, -1, -1):                           # heapify = _siftup from the last parent to the root
            sift_path_form(b, t, b[t])
        assert a == b
        for x in items[k:]:
            if x[0] > a[0][0]:                                        # find_k_largest: strictly greater than the root's score
                heapq.heapreplace(a, x)
                sift_path_form(b, 0, x)
                assert a == b
