#!/usr/bin/env python3
"""Upper bound for a cache-blocked SpMM: time the segmented kernel when the gathered rows are folded into
a window of W rows (same nnz, same row lengths, same arithmetic; only the gather footprint changes)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import SpmmPlan, joint_norm_adjacency
from qrec_amd.synth import make_dataset
capi.init(0)
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; N = nu + ni
indptr, indices, values = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
X = DB.from_numpy(np.random.default_rng(0).standard_normal((N, 64)).astype(np.float32)); Y = DB.zeros((N, 64), np.float32)
e0, e1 = capi.Event(), capi.Event()
def t(plan, reps=30):
    ts = []
    for _ in range(reps):
        e0.record(); capi.spmm_csr(plan, X, Y, 64); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
    return float(np.median(ts)) * 1e3
out = {}
for W in (0, 65536, 32768, 16384, 8192, 4096, 1024):
    idx = indices if W == 0 else (indices % W).astype(indices.dtype)
    out[f"window_rows_{W or N}"] = round(t(SpmmPlan(indptr, idx, values, 64)), 1)
print(json.dumps(out))
