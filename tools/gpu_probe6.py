#!/usr/bin/env python3
"""Probe 6: item-major Hogwild schedule -- speed, 1-group exactness, paired parity (12 epochs)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c as O
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))
capi.init(0)
d = make_dataset("yelp2018"); U, I, dim = d["n_users"], d["n_items"], 64
indptr, ind = to_csr(U, d["train_u"], d["train_i"]); u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
n = ind.size
perm = np.argsort(ind, kind="stable"); u_s = np.ascontiguousarray(u[perm]); i_s = np.ascontiguousarray(ind[perm])
rng = np.random.default_rng(3)
P0 = (rng.random((U, dim)) / 3).astype(np.float32); Q0 = (rng.random((I, dim)) / 3).astype(np.float32)
t = DeviceTables(P0, Q0, np.float32)
sgd = BprSgd(t, u_s, i_s, CSR(indptr, ind))     # item-major triplet list; sampler keyed by scheduled position
e0, e1 = capi.Event(), capi.Event()
# speed
sgd.sample_negatives_device(1, 0)
for chunk in (16, 32, 64):
    for fe in (8, 16, 64):
        for groups in (4096, 8192):
            ts = []
            for rep in range(6):
                e0.record(); capi.bpr_sgd_hogwild_item_major(t.P, t.Q, dim, t.ld, sgd.d_u, sgd.d_i, sgd.d_j, n, chunk, groups, fe, 0.01, 0.001, 0.001, sgd.d_stats); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
            print(json.dumps(dict(chunk=chunk, flush_every=fe, groups=groups, ms=float(np.median(ts[1:])))), flush=True)
# 1 epoch deviation from sequential (in the item-major ORDER the sequential result differs from user-major; compare to oracle on the same order)
j = sgd.d_j.numpy()
for lr in (0.01, 0.05):
    Pr, Qr = P0.astype(np.float64), Q0.astype(np.float64); lref = O.bpr_sgd(Pr, Qr, u_s, i_s, j, lr, 0.001, 0.001)
    for fe in (8, 32):
        t.upload(P0, Q0); capi._check(capi.load().qrec_memset(sgd.d_stats.ptr, 0, 8, None))
        capi.bpr_sgd_hogwild_item_major(t.P, t.Q, dim, t.ld, sgd.d_u, sgd.d_i, sgd.d_j, n, 32, 0, fe, lr, 0.001, 0.001, sgd.d_stats)
        Pg, Qg = t.download(); print(json.dumps(dict(lr=lr, flush_every=fe, P=rel(Pg, Pr), Q=rel(Qg, Qr), loss=abs(sgd.loss() - lref) / lref)), flush=True)
# paired 12-epoch run vs the exact-order port in the REFERENCE (user-major) order with the same (u,i)->j pairing
inv = np.empty_like(perm); inv[perm] = np.arange(n)
for lr0, seed in ((0.01, 7), (0.05, 7), (0.05, 9)):
    t.upload(P0, Q0); Pc, Qc = P0.astype(np.float64), Q0.astype(np.float64)
    lr_g = lr_c = lr0; last_g = last_c = 0.0
    for k in range(12):
        sgd.sample_negatives_device(seed, k); js = sgd.d_j.numpy(); j_um = np.ascontiguousarray(js[inv])   # same j for the same (u,i)
        capi._check(capi.load().qrec_memset(sgd.d_stats.ptr, 0, 8, None))
        capi.bpr_sgd_hogwild_item_major(t.P, t.Q, dim, t.ld, sgd.d_u, sgd.d_i, sgd.d_j, n, 32, 0, 8, lr_g, 0.001, 0.001, sgd.d_stats)
        nll, sp, sq = sgd.epoch_stats(); lg = nll + 0.001 * sp + 0.001 * sq
        lc = O.bpr_sgd(Pc, Qc, u, ind, j_um, lr_c, 0.001, 0.001) + 0.001 * O.sumsq(Pc) + 0.001 * O.sumsq(Qc)
        if k > 0:
            lr_g *= 1.05 if abs(last_g) > abs(lg) else 0.5; lr_c *= 1.05 if abs(last_c) > abs(lc) else 0.5
        last_g, last_c = lg, lc
    print(json.dumps(dict(lr0=lr0, seed=seed, final_gpu=lg, final_cpu=lc, lr_gpu=lr_g, lr_cpu=lr_c)), flush=True)
