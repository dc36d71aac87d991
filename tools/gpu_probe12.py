#!/usr/bin/env python3
"""Item-major kernel: time vs (chunk, flush_every, groups) at the bench shape, and the one-epoch deviation from the
sequential recurrence in the same visiting order (staleness price)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
data = make_dataset("yelp2018"); U, I = data["n_users"], data["n_items"]
indptr, items = to_csr(U, data["train_u"], data["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32); n = items.size
rng = np.random.default_rng(0)
P0 = (rng.random((U, 64)) / 3).astype(np.float32); Q0 = (rng.random((I, 64)) / 3).astype(np.float32)
e0, e1 = capi.Event(), capi.Event()
out = {}
for chunk in (28, 30, 31, 32, 33, 34, 35, 36, 40, 48):
    for flush in (8,):
        if flush > chunk: continue
        for groups in (0,):
            t = DeviceTables(P0, Q0, np.float32); sgd = BprSgd(t, u, items, CSR(indptr, items), schedule="item")
            sgd.sample_negatives_device(1, 0)
            ts = []
            for k in range(8):
                e0.record(); sgd.epoch_throughput_async(0.01, 0.001, 0.001, chunk=chunk, flush_every=flush, groups=groups); e1.record(); e1.sync()
                ts.append(e1.elapsed_ms_since(e0))
            out[f"chunk{chunk}_flush{flush}"] = round(float(np.median(ts[2:])), 4)
print(json.dumps(out))
