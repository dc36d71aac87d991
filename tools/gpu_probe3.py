#!/usr/bin/env python3
"""Probe 3: Hogwild throughput and deviation-from-sequential vs number of groups in flight."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.synth import make_dataset, to_csr
from oracle import c as O
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))
capi.init(0)
d = make_dataset("yelp2018"); U, I = d["n_users"], d["n_items"]
indptr, ind = to_csr(U, d["train_u"], d["train_i"]); n = ind.size
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
j = O.bpr_sample_epoch(O.MT.cpython_seed(1), indptr, ind, I)
rng = np.random.default_rng(0); dim = 64
P0 = (rng.random((U, dim)) / 3).astype(np.float32); Q0 = (rng.random((I, dim)) / 3).astype(np.float32)
du, di, dj = DB.from_numpy(u), DB.from_numpy(ind), DB.from_numpy(j); dl = DB.zeros(1, np.float64)
dP, dQ = DB.from_numpy(P0), DB.from_numpy(Q0); e0, e1 = capi.Event(), capi.Event()
for lr in (0.01, 0.05, 0.2):
    Pr, Qr = P0.astype(np.float64), Q0.astype(np.float64); lref = O.bpr_sgd(Pr, Qr, u, ind, j, lr, 0.001, 0.001)
    for chunk in (8, 32):
        for groups in (1024, 2048, 4096, 8192, 16384, 32768):
            dP.upload(P0); dQ.upload(Q0); dl.fill_bytes(0)
            capi.bpr_sgd_hogwild(dP, dQ, dim, dim, du, di, dj, n, chunk, groups, lr, 0.001, 0.001, dl, 3); capi.device_sync()
            res = dict(lr=lr, chunk=chunk, groups=groups, P=rel(dP.numpy(), Pr), Q=rel(dQ.numpy(), Qr), loss=abs(dl.numpy()[0] - lref) / lref)
            ts = []
            for rep in range(5):
                e0.record(); capi.bpr_sgd_hogwild(dP, dQ, dim, dim, du, di, dj, n, chunk, groups, lr, 0.001, 0.001, dl, 3); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
            res["ms"] = float(np.median(ts)); print(json.dumps(res), flush=True)
