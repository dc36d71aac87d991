#!/usr/bin/env python3
"""Probe 9: emulate N data-parallel ranks on one GPU (sequentially): each 'rank' owns its own users (P_r) and
runs one epoch on its replica of Q from the common start; then Q = Q_start + sum_r (Q_r - Q_start) (what the
all-reduce does).  Global loss / bold driver.  Is the rule stable over 35 steps at lr0 = 0.01?"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
d = make_dataset("yelp2018"); U, I, dim = d["n_users"], d["n_items"], 64
indptr, ind = to_csr(U, d["train_u"], d["train_i"]); u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
Q0 = (np.random.default_rng(999).random((I, dim)) / 3).astype(np.float32)
for N, mode in ((1, "sum"), (8, "sum"), (8, "mean")):
    ranks = []
    for r in range(N):
        P0 = (np.random.default_rng(1000 + r).random((U, dim)) / 3).astype(np.float32)
        t = DeviceTables(P0, Q0, np.float32); ranks.append((t, BprSgd(t, u, ind, CSR(indptr, ind), schedule="item")))
    Q = Q0.copy(); lr, last = 0.01, 0.0; hist = []
    for k in range(35):
        nll = sp = 0.0; delta = np.zeros_like(Q, dtype=np.float64)
        for r, (t, s) in enumerate(ranks):
            t.Q.upload(Q); s.sample_negatives_device(2018 + r, k); s.epoch_throughput_async(lr, 0.001, 0.001)
            a, b, _ = s.epoch_stats(); nll += a; sp += b
            delta += t.Q.numpy().astype(np.float64) - Q
        Q = (Q + (delta if mode == "sum" else delta / N)).astype(np.float32)
        loss = nll + 0.001 * sp + 0.001 * float((Q.astype(np.float64) ** 2).sum())
        if k > 0: lr *= 1.05 if abs(last) > abs(loss) else 0.5
        last = loss; hist.append(round(loss / N))
    print(json.dumps(dict(N=N, mode=mode, loss_per_rank=hist[::4] + [hist[-1]], final_lr=lr, finite=bool(np.isfinite(Q).all()), qnorm=float(np.linalg.norm(Q)))), flush=True)
