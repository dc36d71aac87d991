#!/usr/bin/env python3
"""Probe 5: paired comparison -- the SAME per-epoch negatives (device Philox) drive the exact-order
CPU port and the GPU throughput kernel; 12 epochs, bold driver, lr0 = 0.05."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c as O
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
d = make_dataset("yelp2018"); U, I, dim = d["n_users"], d["n_items"], 64
indptr, ind = to_csr(U, d["train_u"], d["train_i"]); u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
rng = np.random.default_rng(3)
P0 = (rng.random((U, dim)) / 3).astype(np.float32); Q0 = (rng.random((I, dim)) / 3).astype(np.float32)
t = DeviceTables(P0, Q0, np.float32); sgd = BprSgd(t, u, ind, CSR(indptr, ind))
for seed in (7, 8, 9):
    t.upload(P0, Q0); Pc, Qc = P0.astype(np.float64), Q0.astype(np.float64)
    lr_g = lr_c = 0.05; last_g = last_c = 0.0; tr_g, tr_c = [], []
    for k in range(12):
        sgd.sample_negatives_device(seed, k); j = sgd.d_j.numpy()
        capi._check(capi.load().qrec_memset(sgd.d_stats.ptr, 0, 8, None))
        capi.bpr_sgd_hogwild(t.P, t.Q, dim, t.ld, sgd.d_u, sgd.d_i, sgd.d_j, sgd.n, 32, 0, lr_g, 0.001, 0.001, sgd.d_stats, 0)
        nll, sp, sq = sgd.epoch_stats(); lg = nll + 0.001 * sp + 0.001 * sq
        lc = O.bpr_sgd(Pc, Qc, u, ind, j, lr_c, 0.001, 0.001) + 0.001 * O.sumsq(Pc) + 0.001 * O.sumsq(Qc)
        if k > 0:
            lr_g *= 1.05 if abs(last_g) > abs(lg) else 0.5
            lr_c *= 1.05 if abs(last_c) > abs(lc) else 0.5
        last_g, last_c = lg, lc; tr_g.append(float(lg)); tr_c.append(float(lc))
    print(json.dumps(dict(seed=seed, gpu=tr_g, cpu=tr_c, lr_gpu=lr_g, lr_cpu=lr_c)), flush=True)
