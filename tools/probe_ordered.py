#!/usr/bin/env python3
"""Per-triplet time of the order-exact BPR kernel: fp64 vs fp32 tables, tables cache-resident vs 1.2 GB (latency-bound?)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
capi.init(0); rng = np.random.default_rng(0); out = {}
for name, U, I in (("yelp", 31668, 38048), ("big", 1_250_000, 1_000_000)):
    n = 300_000
    u = np.sort(rng.integers(0, U, n)).astype(np.int32); i = rng.integers(0, I, n).astype(np.int32); j = rng.integers(0, I, n).astype(np.int32)
    for dt in (np.float64, np.float32):
        P0 = (rng.random((U, 64)) / 3).astype(dt); Q0 = (rng.random((I, 64)) / 3).astype(dt)
        t = DeviceTables(P0, Q0, dt); s = BprSgd(t, u, i); s.set_negatives(j)
        s.epoch_ordered(0.01, 0.001, 0.001); capi.device_sync()
        t0 = time.perf_counter(); s.epoch_ordered(0.01, 0.001, 0.001); capi.device_sync(); dtm = time.perf_counter() - t0
        out[f"{name}_{np.dtype(dt).name}"] = dict(ns_per_triplet=dtm / n * 1e9)
        del t, s
print(json.dumps(out))
