#!/usr/bin/env python3
"""MHCN step time at the shape of the reference's own social dataset (dataset/lastfm: 1,892 users, 17,632 items, 92,834
ratings, 25,434 follow edges -- synthesised with the same counts, the file itself does not travel), config/MHCN.conf
options: 2 layers, d = 50, batch 2000, ss_rate 0.01."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import MHCNTrainer, mhcn_channel_graphs
capi.init(0); rng = np.random.default_rng(0); out = {}
nu, ni, E, Rn, d, B = 1892, 17632, 92834, 25434, 50, 2000
pu = 1.0 / np.arange(1, nu + 1) ** 0.3; pi = 1.0 / np.arange(1, ni + 1) ** 0.8
uid = rng.choice(nu, int(E * 1.2), p=pu / pu.sum()); iid = rng.choice(ni, int(E * 1.2), p=pi / pi.sum())
pairs = np.unique(np.stack([uid, iid], 1), axis=0); pairs = pairs[rng.permutation(len(pairs))[:E]]; uid, iid = pairs[:, 0], pairs[:, 1]
fo = rng.integers(0, nu, Rn); fe = rng.integers(0, nu, Rn); back = rng.random(Rn) < 0.5      # half of the ties are mutual
rel = np.unique(np.concatenate([np.stack([fo, fe], 1), np.stack([fe[back], fo[back]], 1)])[lambda a: a[:, 0] != a[:, 1]] if False else
                np.concatenate([np.stack([fo, fe], 1), np.stack([fe[back], fo[back]], 1)]), axis=0)
rel = rel[rel[:, 0] != rel[:, 1]]
t0 = time.perf_counter(); H, R = mhcn_channel_graphs(nu, ni, uid, iid, np.ones(uid.size), rel[:, 0], rel[:, 1]); out["host_channel_graphs_s"] = time.perf_counter() - t0
out["nnz"] = dict(H_s=int(H[0].nnz), H_j=int(H[1].nnz), H_p=int(H[2].nnz), R=int(R.nnz))
lim = np.sqrt(6.0 / (2 * d)); w = {}
for k in (1, 2, 3, 4):
    for pre in ("gating", "sgating"):
        w[f"{pre}{k}"] = rng.uniform(-lim, lim, (d, d)).astype(np.float32); w[f"{pre}_bias{k}"] = rng.uniform(-0.3, 0.3, (1, d)).astype(np.float32)
w["attention"] = rng.uniform(-0.3, 0.3, (1, d)).astype(np.float32); w["attention_mat"] = rng.uniform(-lim, lim, (d, d)).astype(np.float32)
t0 = time.perf_counter()
tr = MHCNTrainer((rng.standard_normal((nu, d)) * 0.005).astype(np.float32), (rng.standard_normal((ni, d)) * 0.005).astype(np.float32), w, H, R, 2, 0.001, 0.001, 0.01)
out["trainer_setup_s"] = time.perf_counter() - t0
perm = rng.permutation(E); u = uid[perm].astype(np.int32); i = iid[perm].astype(np.int32); j = rng.integers(0, ni, E).astype(np.int32)
du, di, dj = DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j)
nb = E // B
def step(s):
    s %= nb
    tr.train_step_async(du.ptr + 4 * s * B, di.ptr + 4 * s * B, dj.ptr + 4 * s * B, B)
for s in range(5): step(s)
capi.device_sync(); t0 = time.perf_counter()
steps = 60
for s in range(5, 5 + steps): step(s)
capi.device_sync(); dt = (time.perf_counter() - t0) / steps
rec, ss = tr.losses()
out["step"] = dict(ms_per_step=dt * 1e3, triplets_per_s=B / dt, epoch_s=dt * -(-E // B), rec_loss=rec, ss_loss=ss)
print(json.dumps(out))
