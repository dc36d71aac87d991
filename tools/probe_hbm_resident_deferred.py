#!/usr/bin/env python3
"""The HBM-resident slice of config #4 (1.25 M x 1 M, d = 128, 25 M triplets) under the schedule `auto` picks there (item-deferred in sub-epochs):
epoch time by groups in flight, sub-epochs, pass-A chunk, flush interval and item run.  One JSON line per configuration."""
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
j2 = rng.integers(0, I2, n2, dtype=np.int32)
t = DeviceTables(P2, Q2, np.float32)
alg = n2 * B.bytes_per_triplet(d2)
e0, e1 = capi.Event(), capi.Event()
for cfg in (json.load(open(sys.argv[1])) if os.path.exists(sys.argv[1]) else json.loads(sys.argv[1])):
    s = BprSgd(t, u2, i2, None, schedule=cfg.get("schedule", "item-deferred"), sub_epochs=cfg.get("S", 4), sub_chunk=cfg.get("sub_chunk"), item_run=cfg.get("item_run"), fresh=cfg.get("fresh", False))
    s.set_negatives(j2)
    ts = []
    for _ in range(4):
        e0.record(); s.epoch_throughput_async(B.LR0, B.REG_U, B.REG_I, chunk=cfg.get("chunk", 32), groups=cfg.get("groups", 0), flush_every=cfg.get("flush", 16)); e1.record(); e1.sync()
        ts.append(e1.elapsed_ms_since(e0))
    ms = float(np.median(ts[1:]))
    print(json.dumps({**cfg, "ms": round(ms, 3), "frac": round(alg / ms / 1e6 / 8000, 4)}), flush=True)
    del s
