#!/usr/bin/env python3
"""BUIR step (config/BUIR.conf options, Yelp2018 shape, d=50) alone, for rocprofv3 --kernel-trace --stats."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import BUIRTrainer, joint_norm_adjacency
from qrec_amd.synth import make_dataset
capi.init(0); rng = np.random.default_rng(0)
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; nn = d["train_u"].size
perm = rng.permutation(nn)
du, di = DB.from_numpy(d["train_u"][perm].astype(np.int32)), DB.from_numpy(d["train_i"][perm].astype(np.int32))
def sub():
    keep = rng.permutation(nn)[:nn // 2]
    return joint_norm_adjacency(nu, ni, d["train_u"][keep], d["train_i"][keep])
lim = np.sqrt(6 / (nu + 50))
bt = BUIRTrainer(rng.uniform(-lim, lim, (nu, 50)).astype(np.float32), rng.uniform(-lim, lim, (ni, 50)).astype(np.float32),
                 rng.uniform(-0.24, 0.24, (50, 50)).astype(np.float32), rng.uniform(-0.3, 0.3, (1, 50)).astype(np.float32), 2, 0.001, 0.995)
t0 = time.perf_counter(); bt.set_subgraphs(sub(), sub()); t_plan = time.perf_counter() - t0
B = 2000
for k in range(5): bt.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, B)
capi.device_sync(); t0 = time.perf_counter()
for k in range(60): bt.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, B)
t_host = (time.perf_counter() - t0) / 60
capi.device_sync(); dt = (time.perf_counter() - t0) / 60
print(json.dumps(dict(ms_per_step=dt * 1e3, host_enqueue_ms=t_host * 1e3, plans_s=t_plan)))
