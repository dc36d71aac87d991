#!/usr/bin/env python3
"""How much Recall@20 of ORDER-EXACT, SEQUENTIAL, fp64 BPR training (the oracle's restatement of model/ranking/BPR.py:45-53 with
the bold driver of base/iterativeRecommender.py:56-63) moves when nothing changes but the ORDER in which an epoch's triplets are
visited -- no GPU, no Hogwild, no fp32 anywhere.  CPU only (runs in the build container); the yardstick for tools/paired_recall.py:
a throughput schedule cannot be closer to the reference than the sequential statement of its own visiting order is.

orders (same triplets, same negative for every (u, i), same initial tables):
  reference            user-major PositiveSet order (BPR.py:31-34)
  perturbed            reference order, P0 multiplied by (1 + 1e-7 noise): is the measure chaotic at this setting?
  random               a fresh random permutation every epoch
  user-stride-32       chunks of 32 consecutive triplets of the reference order, chunks in golden-ratio stride order (the user-major kernel's grid)
  item-run-R           triplets sorted by positive item, cut into runs of R, runs in golden-ratio stride order; R = 32 is the order of
                       the item-major kernel as shipped in rounds 1-3, R = 8 the round-4 default

usage: order_sensitivity.py <out.json> [dataset ...]"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c as O                                  # noqa: E402
from tools.paired_recall import DIM, MAX_LR, REG, initial_tables, load_dataset      # noqa: E402


def recall20_cpu(P, Q, d, N=20):
    """util/measure.py:106-109 on the host (torch CPU matmul + top-k; rated items masked to 0 as base/recommender.py:147-149)"""
    import torch
    U, I = d["n_users"], d["n_items"]
    users = np.unique(d["test_u"])
    Qt = torch.from_numpy(Q.astype(np.float32))
    test_keys = np.unique(d["test_u"].astype(np.int64) * I + d["test_i"])
    cnt = np.bincount(d["test_u"], minlength=U)[users]
    hits = np.zeros(users.size)
    indptr, ind = d["indptr"], d["items"]
    for s in range(0, users.size, 4096):
        us = users[s:s + 4096]
        sc = torch.from_numpy(P[us].astype(np.float32)) @ Qt.T
        rows = np.repeat(np.arange(us.size), indptr[us + 1] - indptr[us])
        cols = np.concatenate([ind[indptr[x]:indptr[x + 1]] for x in us]) if rows.size else np.zeros(0, np.int64)
        sc[torch.from_numpy(rows), torch.from_numpy(cols.astype(np.int64))] = 0
        ids = torch.topk(sc, N, dim=1).indices.numpy()
        hits[s:s + 4096] = np.isin((us.astype(np.int64)[:, None] * I + ids).ravel(), test_keys).reshape(ids.shape).sum(1)
    return float((hits / cnt).mean())


def stride_runs(base, run):
    n = base.size
    n_runs = -(-n // run)
    stride = max(int(n_runs * 0.6180339887498949), 1)
    while math.gcd(stride, n_runs) != 1:
        stride += 1
    slots = (np.arange(n_runs, dtype=np.int64) * stride) % n_runs
    at = (slots[:, None] * run + np.arange(run)[None, :]).ravel()
    return base[at[at < n]]


def main():
    out_path = sys.argv[1]
    names = sys.argv[2:] or ["yelp2018-clustered", "lastfm"]
    res = {"_what": __doc__.split("\n\n")[0], "cases": []}
    for name in names:
        d = load_dataset(name)
        u, items, n = d["u"], d["items"], d["u"].size
        P0f, Q0f = initial_tables(d, 3)
        P0, Q0 = P0f.astype(np.float64), Q0f.astype(np.float64)
        ident = np.arange(n)
        by_item = np.argsort(items, kind="stable")
        orders = {"reference": lambda k, r: None, "perturbed": lambda k, r: None, "random": lambda k, r: r.permutation(n),
                  "user-stride-32": (lambda p: (lambda k, r: p))(stride_runs(ident, 32))}
        for R in (32, 16, 8):
            orders[f"item-run-{R}"] = (lambda p: (lambda k, r: p))(stride_runs(by_item, R))
        for lr0, epochs, every in ((0.01, 40, 5), (0.05, 20, 5)) if name != "lastfm" else ((0.01, 40, 4), (0.05, 20, 2)):
            mt = O.MT.cpython_seed(7)
            negs = [O.bpr_sample_epoch(mt, d["indptr"], items, d["n_items"]) for _ in range(epochs)]
            curves = {}
            for tag, fn in orders.items():
                P = P0 * (1 + 1e-7 * np.random.default_rng(9).standard_normal(P0.shape)) if tag == "perturbed" else P0.copy()
                Q, lr, last, rng, curve = Q0.copy(), lr0, 0.0, np.random.default_rng(5), {}
                for k in range(epochs):
                    p = fn(k, rng)
                    uu, ii, jj = (u, items, negs[k]) if p is None else (np.ascontiguousarray(u[p]), np.ascontiguousarray(items[p]), np.ascontiguousarray(negs[k][p]))
                    loss = O.bpr_sgd(P, Q, uu, ii, jj, lr, REG, REG) + REG * O.sumsq(P) + REG * O.sumsq(Q)
                    if k > 0:
                        lr *= 1.05 if abs(last) > abs(loss) else 0.5
                    lr = min(lr, MAX_LR); last = loss
                    if (k + 1) % every == 0 or k + 1 == epochs:
                        curve[k + 1] = recall20_cpu(P, Q, d)
                curves[tag] = curve
                print(name, lr0, tag, {m: round(v, 5) for m, v in curve.items()}, flush=True)
            ref = curves["reference"]
            peak = max(ref, key=lambda m: ref[m])
            case = {"dataset": name, "lr0": lr0, "epochs": epochs, "dim": DIM, "peak_epoch": peak, "recall_reference_at_peak": ref[peak],
                    "vs_reference": {tag: {"at_peak": abs(c[peak] - ref[peak]), "at_final": abs(c[epochs] - ref[epochs]),
                                           "max_over_marks": max(abs(c[m] - ref[m]) for m in ref)} for tag, c in curves.items() if tag != "reference"},
                    "curves": {tag: [[m, v] for m, v in c.items()] for tag, c in curves.items()}}
            res["cases"].append(case)
            json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
