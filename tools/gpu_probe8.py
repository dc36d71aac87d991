import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import SpmmPlan, joint_norm_adjacency
from qrec_amd.synth import make_dataset
capi.init(0); d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; n = nu + ni
adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"]); rng = np.random.default_rng(0)
X = np.zeros((n, 64), np.float32); nz = rng.permutation(n)[:6000]; X[nz] = 1.0
mask = np.zeros((n + 31) // 32, np.uint32); np.bitwise_or.at(mask, nz >> 5, (np.uint32(1) << (nz & 31).astype(np.uint32)))
dX, dY, dM = DB.from_numpy(X), DB.zeros((n, 64), np.float32), DB.from_numpy(mask); e0, e1 = capi.Event(), capi.Event()
for seg in (32, 64, 128, 256):
    plan = SpmmPlan(adj[0], adj[1], adj[2], 64, seg_len=seg)
    for name, m in (("dense", None), ("masked", dM)):
        ts = []
        for r in range(12):
            e0.record(); capi.spmm_csr(plan, dX, dY, 64, d_addend=dX, addend_scale=1.0, d_x_row_mask=m); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
        print(seg, name, "segs", plan.n_segs, "long", plan.n_long, "ms", round(float(np.median(ts[2:])), 4), flush=True)
