#!/usr/bin/env python3
"""Deferred negatives (bpr_sgd.hip, round 3) against the one-pass item-major kernel at the Yelp2018 shape: epoch time by grid
size, and a first look at what the reordering does to a run (loss after a few epochs from the same start)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
d = make_dataset(os.environ.get("PROBE_SHAPE", "yelp2018")); U, I = d["n_users"], d["n_items"]
indptr, items = to_csr(U, d["train_u"], d["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32); n = items.size
rng = np.random.default_rng(0); P0 = (rng.random((U, 64)) / 3).astype(np.float32); Q0 = (rng.random((I, 64)) / 3).astype(np.float32)
chunk = balanced_chunk(n)
out = {"n": int(n), "chunk": chunk}
e0, e1 = capi.Event(), capi.Event()
for sched in ("item", "item-deferred"):
    t = DeviceTables(P0, Q0, np.float32); s = BprSgd(t, u, items, CSR(indptr, items), schedule=sched)
    for groups in ((0,) if sched == "item" else (0, 4096, 8192, 32768, 65536)):
        ts = []
        for k in range(8):
            s.sample_negatives_device(5, k)
            e0.record(); s.epoch_throughput_async(0.01, 0.001, 0.001, chunk=chunk, groups=groups, flush_every=16); e1.record(); e1.sync()
            ts.append(e1.elapsed_ms_since(e0))
        out[f"{sched}_groups{groups}"] = float(np.median(ts[2:]))
        print(sched, "groups", groups, "ms/epoch (kernels only, incl. the sort)", round(float(np.median(ts[2:])), 4), flush=True)
    # trajectory from the same start: 10 epochs at lr 0.02
    t.upload(P0, Q0); losses = []
    for k in range(10):
        s.sample_negatives_device(9, k); s.epoch_throughput_async(0.02, 0.001, 0.001, chunk=chunk, flush_every=16)
        nll, sp, sq = s.epoch_stats(); losses.append(nll + 0.001 * (sp + sq))
    out[f"{sched}_losses"] = losses
    print(sched, "losses", [round(x) for x in losses], flush=True)
print(json.dumps(out))
