#!/usr/bin/env python3
"""bench.py's roofline_hbm_resident workload (1.15 GB of tables) by number of groups in flight, user- and item-major one-pass kernels:
does a table that lives in HBM want more of the grid than the cache-resident Yelp2018 shape (where 4,096 groups are as fast as 32,768)?"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
capi.init(0)
rng = np.random.default_rng(0)
U2, I2, n2, d2 = 1_250_000, 1_000_000, 25_000_000, 128
u2 = np.sort(rng.integers(0, U2, n2, dtype=np.int32)); i2 = rng.integers(0, I2, n2, dtype=np.int32)
blk = (rng.random((50_000, d2)) / 3).astype(np.float32)
P2 = np.empty((U2, d2), np.float32); Q2 = np.empty((I2, d2), np.float32)
for a in (P2, Q2):
    for k in range(0, a.shape[0], 50_000):
        a[k:k + 50_000] = blk[:min(50_000, a.shape[0] - k)]
jn = rng.integers(0, I2, n2, dtype=np.int32)
out = {}
e0, e1 = capi.Event(), capi.Event()
for sched in ("user", "item"):
    t = DeviceTables(P2, Q2, np.float32); s = BprSgd(t, u2, i2, None, schedule=sched); s.set_negatives(jn)
    for groups in (0, 8192, 16384, 32768, 65536):
        ts = []
        for _ in range(4):
            e0.record(); s.epoch_throughput_async(B.LR0, B.REG_U, B.REG_I, groups=groups); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
        ms = float(np.median(ts[1:])); out[f"{sched}_groups{groups}"] = ms
        print(sched, "groups", groups, round(ms, 2), "ms", round(n2 * B.bytes_per_triplet(d2) / ms / 1e6 / B.HBM_PEAK_GBPS, 4), flush=True)
    del s, t
print(json.dumps(out))
