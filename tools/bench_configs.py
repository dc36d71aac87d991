#!/usr/bin/env python3
"""Kernel timings for the BASELINE.json configs that bench.py does not cover: BPR on the ML-1M shape
(config #2), the HBM-sized slice of config #4 (U=1.25M, I=1M, d=128), NGCF step (config #5)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.graph import NGCFTrainer, joint_norm_adjacency
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
capi.init(0); out = {}; e0, e1 = capi.Event(), capi.Event(); rng = np.random.default_rng(0)
def time_epoch(sgd, reps=8, **kw):
    ts = []
    for r in range(reps):
        e0.record(); sgd.epoch_throughput_async(0.01, 0.001, 0.001, **kw); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
    return float(np.median(ts[1:]))
# config 2: ML-1M shape d=64
d = make_dataset("ml1m"); U, I = d["n_users"], d["n_items"]; indptr, ind = to_csr(U, d["train_u"], d["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32); n = ind.size
P0 = (rng.random((U, 64)) / 3).astype(np.float32); Q0 = (rng.random((I, 64)) / 3).astype(np.float32)
for sched in ("item", "user"):
    t = DeviceTables(P0, Q0, np.float32); s = BprSgd(t, u, ind, CSR(indptr, ind), schedule=sched); s.sample_negatives_device(1, 0)
    ms = time_epoch(s); out[f"bpr_ml1m_{sched}"] = dict(U=U, I=I, n=n, ms=ms, gtps=n / ms / 1e6, alg_TBps=n * 1548 / ms / 1e9, frac=n * 1548 / ms / 1e9 / 8)
# config 4 slice: U=1.25M, I=1M, d=128, 25M triplets (tables 1.15 GB: HBM-resident)
U2, I2, n2 = 1_250_000, 1_000_000, 25_000_000
u2 = np.sort(rng.integers(0, U2, n2, dtype=np.int32)); i2 = rng.integers(0, I2, n2, dtype=np.int32)
P2 = np.empty((U2, 128), np.float32); Q2 = np.empty((I2, 128), np.float32); blk = (rng.random((50_000, 128)) / 3).astype(np.float32)
for a in (P2, Q2):
    for k in range(0, a.shape[0], 50_000): a[k:k + 50_000] = blk[:min(50_000, a.shape[0] - k)]
for sched in ("item", "user"):
    t = DeviceTables(P2, Q2, np.float32); s = BprSgd(t, u2, i2, None, schedule=sched)
    s.set_negatives(rng.integers(0, I2, n2, dtype=np.int32))
    ms = time_epoch(s, reps=4); out[f"bpr_xl_slice_d128_{sched}"] = dict(U=U2, I=I2, n=n2, ms=ms, gtps=n2 / ms / 1e6, alg_TBps=n2 * 3084 / ms / 1e9, frac=n2 * 3084 / ms / 1e9 / 8)
    del t, s
# config 5: NGCF step, Yelp shape d=64 batch 2048
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
lim = np.sqrt(6 / 128); W = [[rng.uniform(-lim, lim, (64, 64)).astype(np.float32) for _ in range(2)] for _ in range(2)]
tr = NGCFTrainer((rng.standard_normal((nu, 64)) * 0.005).astype(np.float32), (rng.standard_normal((ni, 64)) * 0.005).astype(np.float32), W, adj, 0.002, 1e-3)
nn = d["train_u"].size; perm = rng.permutation(nn); B = 2048
du, di, dj = DB.from_numpy(d["train_u"][perm].astype(np.int32)), DB.from_numpy(d["train_i"][perm].astype(np.int32)), DB.from_numpy(rng.integers(0, ni, nn).astype(np.int32))
for k in range(5): tr.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, dj.ptr + 4 * k * B, B)
capi.device_sync(); t0 = time.perf_counter()
for k in range(60): tr.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, dj.ptr + 4 * k * B, B)
capi.device_sync(); dt = (time.perf_counter() - t0) / 60
out["ngcf_step"] = dict(ms_per_step=dt * 1e3, triplets_per_s=B / dt, epoch_s=dt * -(-nn // B), loss=tr.loss())
# BUIR step (config/BUIR.conf options: 2 layers, tau 0.995, drop 0.5, batch 2000), Yelp shape, d=50
from qrec_amd.graph import BUIRTrainer
def sub():
    keep = rng.permutation(nn)[:nn // 2]
    return joint_norm_adjacency(nu, ni, d["train_u"][keep], d["train_i"][keep])
lim = np.sqrt(6 / (nu + 50))
bt = BUIRTrainer(rng.uniform(-lim, lim, (nu, 50)).astype(np.float32), rng.uniform(-lim, lim, (ni, 50)).astype(np.float32),
                 rng.uniform(-0.24, 0.24, (50, 50)).astype(np.float32), rng.uniform(-0.3, 0.3, (1, 50)).astype(np.float32), 2, 0.001, 0.995)
t0 = time.perf_counter(); so, st_ = sub(), sub(); t_sub = time.perf_counter() - t0
t0 = time.perf_counter(); bt.set_subgraphs(so, st_); t_plan = time.perf_counter() - t0
B = 2000
for k in range(5): bt.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, B)
capi.device_sync(); t0 = time.perf_counter()
for k in range(60): bt.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, B)
capi.device_sync(); dt = (time.perf_counter() - t0) / 60
out["buir_step"] = dict(ms_per_step=dt * 1e3, pairs_per_s=B / dt, epoch_s=dt * -(-nn // B), loss=bt.loss(), host_two_subgraphs_s=t_sub, host_plan_upload_s=t_plan)
print(json.dumps(out))
