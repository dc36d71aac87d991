#!/usr/bin/env python3
"""Probe 4: stability of the throughput-mode trajectory (12 epochs, bold driver) at lr0=0.05 vs
groups in flight; several repetitions each (Hogwild is not bit-reproducible)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for lr0 in (0.05, 0.1):
    for groups in (8192, 4096, 2048, 1024):
        res = []
        for rep in range(6):
            t.upload(P0, Q0); lr, last = lr0, 0.0; halvings = 0
            for k in range(12):
                sgd.sample_negatives_device(7 + rep, k)
                capi._check(capi.load().qrec_memset(sgd.d_stats.ptr, 0, 8, None))
                capi.bpr_sgd_hogwild(t.P, t.Q, dim, t.ld, sgd.d_u, sgd.d_i, sgd.d_j, sgd.n, 32, groups, lr, 0.001, 0.001, sgd.d_stats, 0)
                nll, sp, sq = sgd.epoch_stats(); loss = nll + 0.001 * sp + 0.001 * sq
                if k > 0:
                    if abs(last) > abs(loss): lr *= 1.05
                    else: lr *= 0.5; halvings += 1
                lr = min(lr, 1.0); last = loss
            res.append((round(last), halvings))
        print(json.dumps(dict(lr0=lr0, groups=groups, final_loss_and_halvings=res)), flush=True)
