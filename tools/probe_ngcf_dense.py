#!/usr/bin/env python3
"""Times qrec_ngcf_dense_fwd alone (Yelp2018 shape, d=64) -- development probe."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
capi.init(0); rng = np.random.default_rng(0)
N, ld = 69716, 64
E = DB.from_numpy(rng.standard_normal((N, ld)).astype(np.float32)); S = DB.from_numpy(rng.standard_normal((N, ld)).astype(np.float32))
W1 = DB.from_numpy(rng.standard_normal((ld, ld)).astype(np.float32)); W2 = DB.from_numpy(rng.standard_normal((ld, ld)).astype(np.float32))
P = DB.zeros((N, ld), np.float32)
e0, e1 = capi.Event(), capi.Event()
for k in range(5): capi.ngcf_dense_fwd(E, S, W1, W2, N, ld, P)
ts = []
for k in range(30):
    e0.record(); capi.ngcf_dense_fwd(E, S, W1, W2, N, ld, P); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0) * 1e3)
print(json.dumps({"dbg": os.environ.get("QREC_DBG", "0"), "blocks": os.environ.get("QREC_DBG_BLOCKS", "default"), "us_median": float(np.median(ts)), "us_min": float(np.min(ts))}))
