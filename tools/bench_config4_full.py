#!/usr/bin/env python3
"""BASELINE.json config #4 at FULL size on ONE MI355X: BPR d=128, 10 M users x 1 M items, 200 M triplets per epoch
(20 positives per user, uniform items) -- tables 5.63 GB + 2.4 GB of indices resident in HBM.  The 8-GPU run shards the
users; this shows what one GPU's share of the node does with the whole problem (and that 288 GB hold it 30 times over)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk
capi.init(0); rng = np.random.default_rng(4); out = {}
U, I, per_user, d = 10_000_000, 1_000_000, 20, 128
n = U * per_user
t0 = time.perf_counter()
u = np.repeat(np.arange(U, dtype=np.int32), per_user)
i = rng.integers(0, I, n, dtype=np.int32); j = rng.integers(0, I, n, dtype=np.int32)
blk = (rng.random((100_000, d)) / 3).astype(np.float32)
P0 = np.empty((U, d), np.float32); Q0 = np.empty((I, d), np.float32)
for a in (P0, Q0):
    for k in range(0, a.shape[0], 100_000): a[k:k + 100_000] = blk[:min(100_000, a.shape[0] - k)]
out["host_prepare_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
t = DeviceTables(P0, Q0, np.float32); del P0, Q0
from qrec_amd.engine import resolve_schedule
want = sys.argv[1] if len(sys.argv) > 1 else "auto"            # "auto": what engine.resolve_schedule picks at this size
p_update = sys.argv[2] if len(sys.argv) > 2 else "auto"         # P[u]: "auto" (engine.resolve_p_update: load + store at this collision density), "atomic", "rmw"
sched, _ = resolve_schedule(n, None, want)
s = BprSgd(t, u, i, None, schedule=sched, p_update=p_update)
out["schedule"], out["p_update"], out["collision_density"] = sched, s.p_update, s.collision
s.set_negatives(j)
capi.device_sync(); out["upload_s"] = time.perf_counter() - t0
e0, e1 = capi.Event(), capi.Event(); ts, losses = [], []
chunk = balanced_chunk(n)
for r in range(4):
    e0.record(); s.epoch_throughput_async(0.01, 0.001, 0.001, chunk=chunk); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0)); losses.append(s.loss())
ms = float(np.median(ts[1:]))
out["epoch"] = dict(U=U, I=I, d=d, triplets=n, chunk=chunk, ms_all=ts, ms=ms, g_triplets_per_s=n / ms / 1e6, alg_TBps=n * 3084 / ms / 1e9,
                    frac_of_8TBps=n * 3084 / ms / 1e9 / 8, table_bytes=(U + I) * d * 4, losses=losses)
print(json.dumps(out))
