"""TEST INFRASTRUCTURE -- not part of the product (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package).

The reference's BPR epoch restated at the level the reference itself runs at: one Python-level iteration per triplet, a handful of
numpy vector statements on rows of P and Q, negatives drawn with ``random.choice`` and redrawn while the user rated them
(model/ranking/BPR.py:28-40 sampler + loop, :45-53 the seven statements of ``optimization``, util/qmath.py:127-128 sigmoid).
oracle/qrec_oracle.c states the same arithmetic in C (14 ns per triplet); this module exists for ONE number: what the
interpreter-bound original costs per triplet on the host a bench runs on -- the reference itself cannot travel to the GPU box.
Pinned by tests/test_oracle_golden.py: tables and loss equal the C restatement's to 1e-12 and the drawn negatives are the same
stream."""
from __future__ import annotations

import math
import random

import numpy as np


def bpr_epoch(P: np.ndarray, Q: np.ndarray, indptr: np.ndarray, items: np.ndarray, n_items: int, lr: float, reg_u: float,
              reg_i: float, rng: random.Random | None = None, max_triplets: int | None = None):
    """Train P, Q (float64, in place) on the users' positives in CSR order; returns (sum of -log sigmoid, triplets done, negatives).
    ``rng`` = the generator ``random.choice`` uses (default: the global one, like the reference); ``max_triplets`` stops early
    (timing a bounded sample)."""
    draw = (rng or random).choice
    catalogue = list(range(n_items))
    nll, done, negs = 0.0, 0, []
    for u in range(indptr.size - 1):
        mine = items[indptr[u]:indptr[u + 1]].tolist()
        rated = set(mine)
        for i in mine:
            j = draw(catalogue)
            while j in rated:
                j = draw(catalogue)
            negs.append(j)
            pu = P[u]
            s = 1 / (1 + math.exp(-(pu.dot(Q[i]) - pu.dot(Q[j]))))
            g = lr * (1 - s)
            P[u] += g * (Q[i] - Q[j])
            Q[i] += g * P[u]
            Q[j] -= g * P[u]
            P[u] -= lr * reg_u * P[u]
            Q[i] -= lr * reg_i * Q[i]
            Q[j] -= lr * reg_i * Q[j]
            nll += -math.log(s)
            done += 1
            if max_triplets is not None and done >= max_triplets:
                return nll, done, np.asarray(negs, np.int32)
    return nll, done, np.asarray(negs, np.int32)
