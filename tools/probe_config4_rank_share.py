#!/usr/bin/env python3
"""One rank's share of BASELINE config #4 on 8 GPUs (1.25 M users x 1 M items, d = 128, 25 M triplets per epoch: the HBM-resident slice) run
the way the replicated layout runs it by default -- 8 reconciliation batches per epoch through an identity communicator (no links) -- under the
item-major kernel with P[u] written by atomic deltas and by load + store (engine.resolve_p_update, round 6).  One JSON line per setting."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from qrec_amd import capi, dist as qd
from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk


class NoLinks:
    def __init__(self, world): self.world, self.rank = world, 0
    def allreduce(self, *a, **k): pass
    def allreduce_pair(self, *a, **k): pass


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
chunk = balanced_chunk(n2)
for sched, K in (("item/atomic", 1), ("item/atomic", 8), ("item/rmw", 8), ("item/rmw", 1)):
    s = BprSgd(t, u2, i2, None, schedule="item", batches=K, chunk=chunk, p_update=sched.split("/")[1])
    s.set_negatives(j2)
    step = qd.ReplicatedStep(NoLinks(8), qd.ReplicatedTableSync(NoLinks(8), t.Q))
    stream = capi.Stream()
    s.start_device_driver(0.01, log_capacity=16)
    capi.device_sync()
    def run(m):
        for _ in range(m):
            s.epoch_device_async(B.REG_U, B.REG_I, B.MAX_LR, tol=0.0, chunk=chunk, flush_every=16, stream=stream, dist=step)
    run(1); stream.sync()
    t0 = time.perf_counter(); run(3); stream.sync(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"schedule": sched, "reconciliations_per_epoch": K, "ms_per_epoch_no_links": round(ms, 3), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4),
                      "loss": float(s.driver_log()[-1, 0])}), flush=True)
    del s, step
