#!/usr/bin/env python3
"""Probe 2: Hogwild kernel (contiguous-segment layout) -- 1-group exactness, throughput and
staleness at Yelp2018 shape, XL slice.  Development tool."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.synth import make_dataset, to_csr
from oracle import c as O
OUT = {}
def log(*a): print(*a, flush=True)
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))
def dataset(shape):
    d = make_dataset(shape); indptr, ind = to_csr(d["n_users"], d["train_u"], d["train_i"])
    u = np.repeat(np.arange(d["n_users"], dtype=np.int32), np.diff(indptr)).astype(np.int32)
    return d, indptr, ind, u
def pad(a, ld):
    out = np.zeros((a.shape[0], ld), dtype=a.dtype); out[:, :a.shape[1]] = a; return out
capi.init(0)
rng = np.random.default_rng(0)
d, indptr, ind, u = dataset("small"); U, I, n = d["n_users"], d["n_items"], ind.size
j = O.bpr_sample_epoch(O.MT.cpython_seed(5), indptr, ind, I)
du, di, dj = DB.from_numpy(u), DB.from_numpy(ind), DB.from_numpy(j); dl = DB.zeros(1, np.float64)
for dim in (64, 50, 128, 8, 200):
    ld = {8: 32, 50: 64, 64: 64, 128: 128, 200: 256}[dim]
    P0 = rng.random((U, dim)) / 3; Q0 = rng.random((I, dim)) / 3; lr, ru, ri = 0.05, 0.01, 0.02
    Pr, Qr = P0.copy(), Q0.copy(); lref = O.bpr_sgd(Pr, Qr, u, ind, j, lr, ru, ri)
    P32, Q32 = P0.astype(np.float32), Q0.astype(np.float32); r = {}
    for var in (1, 2, 3, 4):
        for chunk, groups in ((64, 1), (7, 1), (16, 0)):
            dP, dQ = DB.from_numpy(pad(P32, ld)), DB.from_numpy(pad(Q32, ld)); dl.fill_bytes(0)
            capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, chunk, groups, lr, ru, ri, dl, var); capi.device_sync()
            Pg, Qg, lg = dP.numpy(), dQ.numpy(), dl.numpy()[0]
            r[f"v{var}_c{chunk}_g{groups}"] = dict(P=rel(Pg[:, :dim], Pr), Q=rel(Qg[:, :dim], Qr), loss=abs(lg - lref) / lref,
                                                   pad0=bool((Pg[:, dim:] == 0).all() and (Qg[:, dim:] == 0).all()))
    log(dim, json.dumps(r)); OUT[f"small_d{dim}"] = r

d, indptr, ind, u = dataset("yelp2018"); U, I, n = d["n_users"], d["n_items"], ind.size
j = O.bpr_sample_epoch(O.MT.cpython_seed(1), indptr, ind, I)
dim = ld = 64
P0 = (rng.random((U, dim)) / 3).astype(np.float32); Q0 = (rng.random((I, dim)) / 3).astype(np.float32)
lr, ru, ri = 0.01, 0.001, 0.001
Pr, Qr = P0.astype(np.float64), Q0.astype(np.float64); lref = O.bpr_sgd(Pr, Qr, u, ind, j, lr, ru, ri)
du, di, dj = DB.from_numpy(u), DB.from_numpy(ind), DB.from_numpy(j)
dP, dQ = DB.from_numpy(P0), DB.from_numpy(Q0); e0, e1 = capi.Event(), capi.Event()
OUT["yelp"] = {}
for var in (3, 4, 1, 2):
    for chunk in (4, 8, 16, 32, 64):
        dP.upload(P0); dQ.upload(Q0); dl.fill_bytes(0)
        capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, chunk, 0, lr, ru, ri, dl, var); capi.device_sync()
        res = dict(P=rel(dP.numpy(), Pr), Q=rel(dQ.numpy(), Qr), loss=abs(dl.numpy()[0] - lref) / lref)
        times = []
        for rep in range(9):
            e0.record(); capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, chunk, 0, lr, ru, ri, dl, var); e1.record(); e1.sync()
            times.append(e1.elapsed_ms_since(e0))
        res["ms"] = float(np.median(times)); res["gtps"] = n / res["ms"] / 1e6; res["alg_TBps"] = n * 1548 / res["ms"] / 1e9
        OUT["yelp"][f"v{var}_c{chunk}"] = res; log(f"yelp v{var} c{chunk}", json.dumps(res))
try:
    U2, I2, dim2, n2 = 1_250_000, 1_000_000, 128, 25_000_000
    r2 = np.random.default_rng(3)
    u2 = np.sort(r2.integers(0, U2, n2, dtype=np.int32)); i2 = r2.integers(0, I2, n2, dtype=np.int32); j2 = r2.integers(0, I2, n2, dtype=np.int32)
    dP2 = DB((U2, dim2), np.float32); dQ2 = DB((I2, dim2), np.float32)
    blk = (r2.random((250_000, dim2)) / 3).astype(np.float32)
    for tab, rows_ in ((dP2, U2), (dQ2, I2)):
        for s in range(0, rows_, 250_000):
            capi._check(capi.load().qrec_memcpy_h2d(tab.ptr + s * dim2 * 4, blk.ctypes.data, min(250_000, rows_ - s) * dim2 * 4, None))
    du2, di2, dj2 = DB.from_numpy(u2), DB.from_numpy(i2), DB.from_numpy(j2); OUT["xl"] = {}
    for var in (3, 4, 1):
        for chunk in (8, 16, 32):
            times = []
            for rep in range(3):
                e0.record(); capi.bpr_sgd_hogwild(dP2, dQ2, dim2, dim2, du2, di2, dj2, n2, chunk, 0, 0.01, 0.001, 0.001, dl, var); e1.record(); e1.sync()
                times.append(e1.elapsed_ms_since(e0))
            ms = float(np.median(times)); OUT["xl"][f"v{var}_c{chunk}"] = dict(ms=ms, gtps=n2 / ms / 1e6, alg_TBps=n2 * 3084 / ms / 1e9)
            log("xl", var, chunk, OUT["xl"][f"v{var}_c{chunk}"])
except Exception as ex:
    log("xl failed", repr(ex))
os.makedirs("gpurun_out", exist_ok=True); json.dump(OUT, open("gpurun_out/probe2.json", "w"), indent=1); log("done")
