#!/usr/bin/env python3
"""How far the throughput schedules drift from order-exact training over a LONG run at BPR.conf's settings (lr0 0.01, bold driver,
-max 1): Recall@20 after 25 / 50 / 100 epochs, paired design of bench.py's recall check (same negatives for the same (u, i)).
usage: probe_deferred_long.py <schedule> <cpu-cache.npz>   (the order-exact CPU side is computed once and cached)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
schedule, cache = sys.argv[1], sys.argv[2]
MARKS = (25, 50, 100)
capi.init(0)
data = make_dataset("yelp2018"); U, I = data["n_users"], data["n_items"]
indptr, items = to_csr(U, data["train_u"], data["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32); n = items.size
rng = np.random.default_rng(1000); P0 = (rng.random((U, 64)) / 3).astype(np.float32); Q0 = (np.random.default_rng(999).random((I, 64)) / 3).astype(np.float32)
chunk = balanced_chunk(n)
t = DeviceTables(P0, Q0, np.float32); s = BprSgd(t, u, items, CSR(indptr, items), schedule=schedule, n_items=I, chunk=chunk)
s.start_device_driver(B.LR0, log_capacity=max(MARKS))
gpu = {}
for k in range(max(MARKS)):
    s.sample_negatives_device(B.SEED, k)
    s.epoch_device_async(B.REG_U, B.REG_I, B.MAX_LR, tol=0.0, chunk=chunk, flush_every=B.FLUSH_EVERY)
    if k + 1 in MARKS:
        capi.device_sync(); Pg, Qg = t.download(np.float32)
        gpu[k + 1] = (B.evaluate_recall(Pg, Qg, data, indptr, items), float(s.driver_log()[k, 0]), float(s.driver_state()["lr"]))
if not os.path.exists(cache):
    from oracle import c as O
    P, Q = P0.astype(np.float64), Q0.astype(np.float64); lr, last = B.LR0, 0.0; cpu = {}
    for k in range(max(MARKS)):
        s.sample_negatives_device(B.SEED, k); j = s.negatives_reference_order()
        loss = O.bpr_sgd(P, Q, u, items, j, lr, B.REG_U, B.REG_I) + B.REG_U * O.sumsq(P) + B.REG_I * O.sumsq(Q)
        if k > 0:
            lr *= 1.05 if abs(last) > abs(loss) else 0.5
        lr = min(lr, B.MAX_LR); last = loss
        if k + 1 in MARKS:
            cpu[k + 1] = (B.evaluate_recall(P, Q, data, indptr, items), loss, lr)
    np.savez(cache, marks=np.array(MARKS), vals=np.array([cpu[m] for m in MARKS]))
z = np.load(cache); cpu = {int(m): tuple(v) for m, v in zip(z["marks"], z["vals"])}
out = {"schedule": schedule, "fresh": os.environ.get("QREC_DEFERRED_FRESH", "0")}
for m in MARKS:
    out[f"epochs_{m}"] = {"recall_gpu": gpu[m][0], "recall_exact_order": cpu[m][0], "abs_diff": abs(gpu[m][0] - cpu[m][0]),
                          "loss_gpu": gpu[m][1], "loss_exact_order": cpu[m][1], "lr_gpu": gpu[m][2], "lr_exact_order": cpu[m][2]}
print(json.dumps(out))
