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


# ---- SBPR (model/ranking/SBPR.py:31-78, numpy path) --------------------------------------------------------------------
# The reference's file raises TypeError at :46 (`self.FPSet[user][kItems]`: a dict indexed with a list) for the first user who
# has social feedback; this restates the loop with that subscript read as `item_k` -- the count the drawn friend-consumed item
# carries -- and everything else as written.  Pinned by tests/golden/sbpr_filmtrust.npz (tests/golden/gen_golden.py
# case_sbpr_filmtrust: the reference's own source with that one token replaced, run here).
def sbpr_sample_epoch(rng: random.Random, ps_users, pos_indptr, pos_items, fp_indptr, fp_items, fp_counts, n_items: int,
                      item_key_user, is_key):
    """The draws of one epoch (SBPR.py:37-55,69-72) -> int32 [n, 5] rows (u, i, k or -1, j, Suk).
    ``ps_users``: user ids in PositiveSet's key order (:38).  Per positive: with social feedback `choice(kItems)` (:44) then the
    negative (:51-53); without, the negative only (:69-71).  The negative is redrawn while it is one of the user's positives OR -- for
    users with social feedback only -- `item_j in self.FPSet` (:52) -- the ITEM's name looked up among the USER names that are keys of the defaultdict FPSet:
    ``item_key_user[j]`` = id of the user whose name equals item j's name (-1: none), ``is_key[user]`` (uint8, updated in place) =
    that user is a key by now (initModel :26 made the social ones keys, the loop's `self.FPSet[user]` :40 makes every visited one)."""
    catalogue = list(range(n_items))
    out = []
    for u in ps_users:
        is_key[u] = 1                                                     # kItems = list(self.FPSet[user].keys())  (:40)
        mine = pos_items[pos_indptr[u]:pos_indptr[u + 1]].tolist()
        rated = set(mine)
        k_items = fp_items[fp_indptr[u]:fp_indptr[u + 1]].tolist()
        k_cnt = fp_counts[fp_indptr[u]:fp_indptr[u + 1]].tolist()
        positions = list(range(len(k_items)))
        for i in mine:
            k, w = -1, 0
            if k_items:
                at = rng.choice(positions)
                k, w = k_items[at], k_cnt[at]
            j = rng.choice(catalogue)
            while j in rated or (k_items and item_key_user[j] >= 0 and is_key[item_key_user[j]]):      # (:70: positives only without feedback)
                j = rng.choice(catalogue)
            out.append((u, i, k, j, w))
    return np.asarray(out, dtype=np.int32).reshape(-1, 5)


def sbpr_epoch(P: np.ndarray, Q: np.ndarray, b: np.ndarray, ps_users, stream: np.ndarray, lr: float, reg_u: float, reg_i: float) -> float:
    """The updates of one epoch over the drawn rows (SBPR.py:41-73), P and Q float64 in place; returns the epoch's loss
    (:57-58, :73 and the per-user terms of :74, which sit inside the user loop)."""
    sig = lambda x: 1 / (1 + math.exp(-x))
    loss = 0.0
    by_user = {}
    for row in stream.tolist():
        by_user.setdefault(row[0], []).append(row)
    for u in ps_users:
        for _, i, k, j, suk in by_user.get(int(u), ()):
            if k >= 0:
                s = sig((P[u].dot(Q[i]) - P[u].dot(Q[k]) + b[i] - b[k]) / (suk + 1))
                P[u] += 1 / (suk + 1) * lr * (1 - s) * (Q[i] - Q[k])
                Q[i] += 1 / (suk + 1) * lr * (1 - s) * P[u]
                Q[k] -= 1 / (suk + 1) * lr * (1 - s) * P[u]
                s = sig(P[u].dot(Q[k]) - P[u].dot(Q[j]) + b[k] - b[j])
                P[u] += lr * (1 - s) * (Q[k] - Q[j])
                Q[k] += lr * (1 - s) * P[u]
                Q[j] -= lr * (1 - s) * P[u]
                P[u] -= lr * reg_u * P[u]
                Q[i] -= lr * reg_i * Q[i]
                Q[j] -= lr * reg_i * Q[j]
                Q[k] -= lr * reg_i * Q[k]
                loss += -math.log(sig((P[u].dot(Q[i]) - P[u].dot(Q[k])) / (suk + 1))) - math.log(sig(P[u].dot(Q[k]) - P[u].dot(Q[j])))
            else:
                s = sig(P[u].dot(Q[i]) - P[u].dot(Q[j]) + b[i] - b[j])
                P[u] += lr * (1 - s) * (Q[i] - Q[j])
                Q[i] += lr * (1 - s) * P[u]
                Q[j] -= lr * (1 - s) * P[u]
                loss += -math.log(s)
        loss += reg_u * (P * P).sum() + reg_i * (Q * Q).sum() + b.dot(b)
    return loss
