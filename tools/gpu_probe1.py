#!/usr/bin/env python3
"""First-contact GPU probe: correctness of every kernel vs the oracle on small inputs, then
throughput + staleness of the Hogwild memory-policy variants at Yelp2018 shape.
Writes gpurun_out/probe1.json.  (Development tool, not part of the product or the tests.)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.synth import make_dataset, to_csr
from oracle import c as O

OUT = {}
def log(*a):
    print(*a, flush=True)

def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))

def dataset(shape):
    d = make_dataset(shape)
    indptr, ind = to_csr(d["n_users"], d["train_u"], d["train_i"])
    u = np.repeat(np.arange(d["n_users"], dtype=np.int32), np.diff(indptr)).astype(np.int32)
    return d, indptr, ind, u

def pad(a, ld):
    out = np.zeros((a.shape[0], ld), dtype=a.dtype); out[:, :a.shape[1]] = a; return out

capi.init(0)
log("device", capi.device_info())
OUT["device"] = capi.device_info()

# ---------------- A. correctness on 'small' ----------------
d, indptr, ind, u = dataset("small")
U, I, n = d["n_users"], d["n_items"], ind.size
mt = O.MT.cpython_seed(5)
j = O.bpr_sample_epoch(mt, indptr, ind, I)
rng = np.random.default_rng(0)
for dim in (64, 50, 128, 8):
    ld = {8: 32, 50: 64, 64: 64, 128: 128}[dim]
    P0 = rng.random((U, dim)) / 3; Q0 = rng.random((I, dim)) / 3
    lr, ru, ri = 0.05, 0.01, 0.02
    # fp64 ordered
    Pr, Qr = P0.copy(), Q0.copy(); lref = O.bpr_sgd(Pr, Qr, u, ind, j, lr, ru, ri)
    dP, dQ = DB.from_numpy(pad(P0, ld)), DB.from_numpy(pad(Q0, ld))
    du, di, dj = DB.from_numpy(u), DB.from_numpy(ind), DB.from_numpy(j)
    dl = DB.zeros(1, np.float64)
    t = time.time(); capi.bpr_sgd_ordered(dP, dQ, capi.F64, dim, ld, du, di, dj, n, lr, ru, ri, dl); capi.device_sync(); dt = time.time() - t
    Pg, Qg, lg = dP.numpy()[:, :dim], dQ.numpy()[:, :dim], dl.numpy()[0]
    r = dict(dim=dim, ordered_f64=dict(P=rel(Pg, Pr), Q=rel(Qg, Qr), loss=abs(lg - lref) / lref, ms=dt * 1e3,
                                        pad_zero=bool((dP.numpy()[:, dim:] == 0).all())))
    # fp32 ordered vs fp32 oracle and fp64 oracle
    P32, Q32 = P0.astype(np.float32), Q0.astype(np.float32)
    Pr32, Qr32 = P32.copy(), Q32.copy(); lref32 = O.bpr_sgd(Pr32, Qr32, u, ind, j, lr, ru, ri)
    dP, dQ = DB.from_numpy(pad(P32, ld)), DB.from_numpy(pad(Q32, ld))
    t = time.time(); capi.bpr_sgd_ordered(dP, dQ, capi.F32, dim, ld, du, di, dj, n, lr, ru, ri, dl); capi.device_sync(); dt = time.time() - t
    Pg, Qg, lg = dP.numpy()[:, :dim], dQ.numpy()[:, :dim], dl.numpy()[0]
    r["ordered_f32"] = dict(P_vs_f32=rel(Pg, Pr32.astype(np.float64)), Q_vs_f32=rel(Qg, Qr32.astype(np.float64)),
                            P_vs_f64=rel(Pg, Pr), Q_vs_f64=rel(Qg, Qr), loss=abs(lg - lref) / lref, ms=dt * 1e3)
    # hogwild variants with ONE group == sequential
    r["hogwild_1group"] = {}
    for var in (1, 2, 3, 4):
        for chunk in (64, 7):
            dP, dQ = DB.from_numpy(pad(P32, ld)), DB.from_numpy(pad(Q32, ld)); dl.fill_bytes(0)
            capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, chunk, 1, lr, ru, ri, dl, var); capi.device_sync()
            Pg, Qg, lg = dP.numpy()[:, :dim], dQ.numpy()[:, :dim], dl.numpy()[0]
            r["hogwild_1group"][f"v{var}_c{chunk}"] = dict(P=rel(Pg, Pr), Q=rel(Qg, Qr), loss=abs(lg - lref) / lref)
    # hogwild full grid: deviation from sequential (staleness) + lr=0 linearity check
    r["hogwild_full"] = {}
    for var in (1, 2, 3, 4):
        dP, dQ = DB.from_numpy(pad(P32, ld)), DB.from_numpy(pad(Q32, ld)); dl.fill_bytes(0)
        capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, 16, 0, lr, ru, ri, dl, var); capi.device_sync()
        Pg, Qg, lg = dP.numpy()[:, :dim], dQ.numpy()[:, :dim], dl.numpy()[0]
        r["hogwild_full"][f"v{var}"] = dict(P=rel(Pg, Pr), Q=rel(Qg, Qr), loss=abs(lg - lref) / lref)
    dP, dQ = DB.from_numpy(pad(P32, ld)), DB.from_numpy(pad(Q32, ld)); dl.fill_bytes(0)
    capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, 16, 0, 0.0, ru, ri, dl, 4); capi.device_sync()
    Pz, Qz = P0.copy(), Q0.copy(); lz = O.bpr_sgd(Pz, Qz, u, ind, j, 0.0, ru, ri)
    r["lr0"] = dict(unchanged=bool((dP.numpy()[:, :dim] == P32).all() and (dQ.numpy()[:, :dim] == Q32).all()),
                    loss=abs(dl.numpy()[0] - lz) / lz)
    # sumsq
    dS = DB.zeros(1, np.float64); dT = DB.from_numpy(pad(P0, ld)); capi.sumsq(dT, capi.F64, U, dim, ld, dS)
    r["sumsq"] = abs(dS.numpy()[0] - (P0 * P0).sum()) / (P0 * P0).sum()
    log(json.dumps(r)); OUT[f"small_d{dim}"] = r

# philox sampler properties
srt = to_csr(U, np.repeat(np.arange(U), np.diff(indptr)), ind)  # not sorted within row yet
rows = np.repeat(np.arange(U, dtype=np.int64), np.diff(indptr))
order = np.lexsort((ind, rows)); ind_sorted = ind[order].astype(np.int32)
dip, dis, dru = DB.from_numpy(indptr), DB.from_numpy(ind_sorted), DB.from_numpy(u)
djo = DB(n, np.int32)
capi.philox_bpr_sample(dip, dis, dru, n, I, 1234, 0, djo); j0 = djo.numpy()
capi.philox_bpr_sample(dip, dis, dru, n, I, 1234, 0, djo); j0b = djo.numpy()
capi.philox_bpr_sample(dip, dis, dru, n, I, 1234, 1, djo); j1 = djo.numpy()
key_pos = set((rows * I + ind).tolist())
viol = sum(1 for a, b in zip(u.tolist(), j0.tolist()) if a * I + b in key_pos)
OUT["philox"] = dict(in_range=bool((j0 >= 0).all() and (j0 < I).all()), positives_hit=viol, deterministic=bool((j0 == j0b).all()),
                     epoch_differs=float((j0 != j1).mean()), mean=float(j0.mean()), expect_mean=(I - 1) / 2)
log("philox", OUT["philox"])

# ---------------- B. throughput at Yelp2018 shape ----------------
d, indptr, ind, u = dataset("yelp2018")
U, I, n = d["n_users"], d["n_items"], ind.size
mt = O.MT.cpython_seed(1); j = O.bpr_sample_epoch(mt, indptr, ind, I)
dim = ld = 64
P0 = (rng.random((U, dim)) / 3).astype(np.float32); Q0 = (rng.random((I, dim)) / 3).astype(np.float32)
lr, ru, ri = 0.01, 0.001, 0.001
Pr, Qr = P0.astype(np.float64), Q0.astype(np.float64)
t = time.time(); lref = O.bpr_sgd(Pr, Qr, u, ind, j, lr, ru, ri); t_or = time.time() - t
log("oracle f64 epoch s", t_or, "triplets/s", n / t_or)
OUT["yelp"] = dict(n=n, U=U, I=I, oracle_f64_s=t_or, variants={})
du, di, dj = DB.from_numpy(u), DB.from_numpy(ind), DB.from_numpy(j)
dl = DB.zeros(1, np.float64)
e0, e1 = capi.Event(), capi.Event()
# ordered fp32 (one epoch)
dP, dQ = DB.from_numpy(P0), DB.from_numpy(Q0)
e0.record(); capi.bpr_sgd_ordered(dP, dQ, capi.F32, dim, ld, du, di, dj, n, lr, ru, ri, dl); e1.record(); e1.sync()
ms = e1.elapsed_ms_since(e0)
OUT["yelp"]["ordered_f32"] = dict(ms=ms, mtps=n / ms / 1e3, P=rel(dP.numpy(), Pr), Q=rel(dQ.numpy(), Qr), loss=abs(dl.numpy()[0] - lref) / lref)
log("ordered_f32", OUT["yelp"]["ordered_f32"])
Pd, Qd = DB.from_numpy(P0.astype(np.float64)), DB.from_numpy(Q0.astype(np.float64))
e0.record(); capi.bpr_sgd_ordered(Pd, Qd, capi.F64, dim, ld, du, di, dj, n, lr, ru, ri, dl); e1.record(); e1.sync()
ms = e1.elapsed_ms_since(e0)
OUT["yelp"]["ordered_f64"] = dict(ms=ms, mtps=n / ms / 1e3, P=rel(Pd.numpy(), Pr), Q=rel(Qd.numpy(), Qr), loss=abs(dl.numpy()[0] - lref) / lref)
log("ordered_f64", OUT["yelp"]["ordered_f64"])
del Pd, Qd
for var in (1, 2, 3, 4):
    for chunk in (8, 16, 32, 64):
        dP.upload(P0); dQ.upload(Q0); dl.fill_bytes(0)
        capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, chunk, 0, lr, ru, ri, dl, var); capi.device_sync()
        res = dict(P=rel(dP.numpy(), Pr), Q=rel(dQ.numpy(), Qr), loss=abs(dl.numpy()[0] - lref) / lref)
        times = []
        for rep in range(7):
            e0.record(); capi.bpr_sgd_hogwild(dP, dQ, dim, ld, du, di, dj, n, chunk, 0, lr, ru, ri, dl, var); e1.record(); e1.sync()
            times.append(e1.elapsed_ms_since(e0))
        res["ms"] = float(np.median(times)); res["ms_min"] = float(min(times))
        res["gtps"] = n / res["ms"] / 1e6; res["alg_TBps"] = n * 1548 / res["ms"] / 1e9
        res["finite"] = bool(np.isfinite(dQ.numpy()).all())
        OUT["yelp"]["variants"][f"v{var}_c{chunk}"] = res
        log(f"v{var} c{chunk}", json.dumps(res))
# philox sampler throughput
rows = np.repeat(np.arange(U, dtype=np.int64), np.diff(indptr)); order = np.lexsort((ind, rows))
dip, dis = DB.from_numpy(indptr), DB.from_numpy(ind[order].astype(np.int32))
times = []
for rep in range(5):
    e0.record(); capi.philox_bpr_sample(dip, dis, du, n, I, 99, rep, dj); e1.record(); e1.sync(); times.append(e1.elapsed_ms_since(e0))
OUT["yelp"]["philox_ms"] = float(np.median(times)); log("philox ms", times)
times = []
dS = DB.zeros(1, np.float64)
for rep in range(5):
    e0.record(); capi.sumsq(dQ, capi.F32, I, dim, ld, dS); e1.record(); e1.sync(); times.append(e1.elapsed_ms_since(e0))
OUT["yelp"]["sumsq_ms"] = float(np.median(times))

# ---------------- C. HBM-bound slice: U=1.25M, I=1M, d=128 (1.15 GB of tables) ----------------
try:
    U2, I2, dim2, n2 = 1_250_000, 1_000_000, 128, 25_000_000
    r2 = np.random.default_rng(3)
    u2 = np.sort(r2.integers(0, U2, n2, dtype=np.int32)); i2 = r2.integers(0, I2, n2, dtype=np.int32); j2 = r2.integers(0, I2, n2, dtype=np.int32)
    dP2 = DB((U2, dim2), np.float32); dQ2 = DB((I2, dim2), np.float32)
    blk = (r2.random((250_000, dim2)) / 3).astype(np.float32)
    for tab, rows_ in ((dP2, U2), (dQ2, I2)):
        for s in range(0, rows_, 250_000):
            capi._check(capi.load().qrec_memcpy_h2d(tab.ptr + s * dim2 * 4, blk.ctypes.data, min(250_000, rows_ - s) * dim2 * 4, None))
    du2, di2, dj2 = DB.from_numpy(u2), DB.from_numpy(i2), DB.from_numpy(j2)
    OUT["xl_slice"] = {}
    for var in (1, 3, 4, 2):
        for chunk in (16, 32):
            times = []
            for rep in range(3):
                dl.fill_bytes(0)
                e0.record(); capi.bpr_sgd_hogwild(dP2, dQ2, dim2, dim2, du2, di2, dj2, n2, chunk, 0, 0.01, 0.001, 0.001, dl, var); e1.record(); e1.sync()
                times.append(e1.elapsed_ms_since(e0))
            ms = float(np.median(times))
            OUT["xl_slice"][f"v{var}_c{chunk}"] = dict(ms=ms, gtps=n2 / ms / 1e6, alg_TBps=n2 * 3084 / ms / 1e9)
            log("xl", var, chunk, OUT["xl_slice"][f"v{var}_c{chunk}"])
except Exception as ex:  # noqa
    OUT["xl_slice_error"] = repr(ex); log("xl failed", ex)

os.makedirs("gpurun_out", exist_ok=True)
json.dump(OUT, open("gpurun_out/probe1.json", "w"), indent=1)
log("done")
