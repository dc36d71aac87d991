#!/usr/bin/env python3
"""Probe 7: paired Recall@20 parity of the item-major schedule (flush interval / groups sweep)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c as O
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.interactions import CSR
from qrec_amd.ranking import DeviceRanker
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
d = make_dataset("yelp2018"); U, I, dim = d["n_users"], d["n_items"], 64
indptr, ind = to_csr(U, d["train_u"], d["train_i"]); u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
n = ind.size
perm = np.argsort(ind, kind="stable"); u_s = np.ascontiguousarray(u[perm]); i_s = np.ascontiguousarray(ind[perm])
inv = np.empty_like(perm); inv[perm] = np.arange(n)
rng = np.random.default_rng(3)
P0 = (rng.random((U, dim)) / 3).astype(np.float32); Q0 = (rng.random((I, dim)) / 3).astype(np.float32)
t = DeviceTables(P0, Q0, np.float32); sgd = BprSgd(t, u_s, i_s, CSR(indptr, ind))
users = np.unique(d["test_u"]).astype(np.int32); test_keys = np.unique(d["test_u"].astype(np.int64) * I + d["test_i"]); cnt = np.bincount(d["test_u"], minlength=U)[users]
def recall(P, Q):
    ids, _ = DeviceRanker(np.ascontiguousarray(P, np.float32), np.ascontiguousarray(Q, np.float32), CSR(indptr, ind)).topk(users, 20)
    return float((np.isin((users.astype(np.int64)[:, None] * I + ids).ravel(), test_keys).reshape(ids.shape).sum(1) / cnt).mean())
cache = {}
for lr0, seed in ((0.01, 7), (0.05, 7), (0.05, 9)):
    for fe, groups in ((8, 4096), (16, 4096), (8, 2048), (32, 4096)):
        t.upload(P0, Q0); lr_g = lr0; last_g = 0.0
        key = (lr0, seed); do_cpu = key not in cache
        if do_cpu: Pc, Qc = P0.astype(np.float64), Q0.astype(np.float64); lr_c = lr0; last_c = 0.0
        for k in range(12):
            sgd.sample_negatives_device(seed, k)
            capi._check(capi.load().qrec_memset(sgd.d_stats.ptr, 0, 8, None))
            capi.bpr_sgd_hogwild_item_major(t.P, t.Q, dim, t.ld, sgd.d_u, sgd.d_i, sgd.d_j, n, 32, groups, fe, lr_g, 0.001, 0.001, sgd.d_stats)
            nll, sp, sq = sgd.epoch_stats(); lg = nll + 0.001 * sp + 0.001 * sq
            if k > 0: lr_g *= 1.05 if abs(last_g) > abs(lg) else 0.5
            last_g = lg
            if do_cpu:
                j_um = np.ascontiguousarray(sgd.d_j.numpy()[inv])
                lc = O.bpr_sgd(Pc, Qc, u, ind, j_um, lr_c, 0.001, 0.001) + 0.001 * O.sumsq(Pc) + 0.001 * O.sumsq(Qc)
                if k > 0: lr_c *= 1.05 if abs(last_c) > abs(lc) else 0.5
                last_c = lc
        if do_cpu: cache[key] = (recall(Pc, Qc), last_c, lr_c)
        Pg, Qg = t.download(np.float32); r_g = recall(Pg, Qg); r_c, l_c, lrc = cache[key]
        print(json.dumps(dict(lr0=lr0, seed=seed, flush_every=fe, groups=groups, recall_gpu=r_g, recall_cpu=r_c, d_recall=r_g - r_c, loss_gap=(last_g - l_c) / l_c, same_lr=abs(lr_g - lrc) < 1e-12)), flush=True)
