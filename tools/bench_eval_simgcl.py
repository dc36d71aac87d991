#!/usr/bin/env python3
"""Timing of (a) the full-rank evaluation pass and (b) the SimGCL training step at the
Yelp2018 shape (config #5 hyper-parameters: L=2, lambda=0.5, eps=0.1, d=64, batch 2048)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import SimGCLTrainer, joint_norm_adjacency, unique_first_appearance
from qrec_amd.interactions import CSR
from qrec_amd.ranking import DeviceRanker
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]
rng = np.random.default_rng(0)
out = {}
# ---- eval: all 31,668 users x 38,048 items, N=20
indptr, ind = to_csr(nu, d["train_u"], d["train_i"])
for dt in (() if "--skip-eval" in sys.argv else (np.float32, np.float64)):
    U = (rng.random((nu, 64)) / 3 - 0.1).astype(dt); V = (rng.random((ni, 64)) / 3 - 0.1).astype(dt)
    rk = DeviceRanker(U, V, CSR(indptr, ind)); users = np.arange(nu, dtype=np.int32)
    rk.topk(users[:2048], 20)
    t0 = time.perf_counter(); ids, sc = rk.topk(users, 20); dtm = time.perf_counter() - t0
    out[f"eval_{np.dtype(dt).name}"] = dict(seconds=dtm, users_per_s=nu / dtm, gflop=2 * nu * ni * 64 / 1e9, tflops_incl_everything=2 * nu * ni * 64 / dtm / 1e12)
# ---- SimGCL step
adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
lim = np.sqrt(6.0 / (nu + 64))
U0 = rng.uniform(-lim, lim, (nu, 64)).astype(np.float32); V0 = rng.uniform(-lim, lim, (ni, 64)).astype(np.float32)
tr = SimGCLTrainer(U0, V0, adj, 2, 0.001, 1e-4, 0.5, 0.1, max_unique=2048)
n = d["train_u"].size; perm = rng.permutation(n); B = 2048
u = d["train_u"][perm].astype(np.int32); i = d["train_i"][perm].astype(np.int32); j = rng.integers(0, ni, n).astype(np.int32)
du, di, dj = DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j)
steps = 60
uu = [unique_first_appearance(u[k * B:(k + 1) * B]) for k in range(steps)]; vv = [(unique_first_appearance(i[k * B:(k + 1) * B]) + nu).astype(np.int32) for k in range(steps)]
duu = [DB.from_numpy(x) for x in uu]; dvv = [DB.from_numpy(x) for x in vv]
def step(k): tr.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, dj.ptr + 4 * k * B, B, duu[k], uu[k].size, dvv[k], vv[k].size)
for k in range(5): step(k)
capi.device_sync(); t0 = time.perf_counter()
for k in range(steps): step(k)
t_host = (time.perf_counter() - t0) / steps          # time to ENQUEUE a step (host side)
capi.device_sync(); dtm = (time.perf_counter() - t0) / steps
out["simgcl"] = dict(ms_per_step=dtm * 1e3, host_enqueue_ms_per_step=t_host * 1e3, triplets_per_s=B / dtm, epoch_s=dtm * -(-n // B), unique_users=int(np.mean([x.size for x in uu])), unique_items=int(np.mean([x.size for x in vv])), losses=tr.losses())
print(json.dumps(out))
