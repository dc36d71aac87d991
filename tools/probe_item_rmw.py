#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 3(b): the item-major throughput kernel with racy read-modify-write row updates instead of atomic deltas
(csrc/bpr_sgd.hip, template parameter RMW; selected by the environment variable QREC_ITEM_RMW = 0 atomic / 1 P[u] by RMW /
3 P[u] and Q[j] by RMW, read once per process).  Prints, for the variant of this process: the epoch time at the Yelp2018 shape and on
the HBM-resident slice of config #4, and how far one epoch lands from the order-exact result of the same triplets (relative, P and Q).

    QREC_ITEM_RMW=1 python tools/probe_item_rmw.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                              # noqa: E402
from oracle import c as O                                      # noqa: E402  (a measurement tool: the checker is allowed here)
from qrec_amd import capi                                      # noqa: E402
from qrec_amd.engine import BprSgd, DeviceTables               # noqa: E402
from qrec_amd.interactions import CSR                          # noqa: E402
from qrec_amd.synth import make_dataset, to_csr                # noqa: E402

capi.init(0)
variant = os.environ.get("QREC_ITEM_RMW", "0")
d = make_dataset("yelp2018")
U, I = d["n_users"], d["n_items"]
indptr, ind = to_csr(U, d["train_u"], d["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
rng = np.random.default_rng(3)
P0 = (rng.random((U, 64)) / 3).astype(np.float32); Q0 = (rng.random((I, 64)) / 3).astype(np.float32)
t = DeviceTables(P0, Q0, np.float32)
s = BprSgd(t, u, ind, CSR(indptr, ind), schedule="item")
s.sample_negatives_device(7, 0)
j = s.negatives_reference_order()
chunk, groups = s.launch_grid(None)
ms = B._time_events(capi, lambda: s.epoch_throughput_async(0.01, 0.001, 0.001, chunk=chunk, groups=groups), 20)
out = {"QREC_ITEM_RMW": variant, "yelp2018_ms_per_epoch": ms, "yelp2018_G_triplets_per_s": u.size / ms / 1e6,
       "yelp2018_roofline_frac": u.size * B.bytes_per_triplet(64) / ms / 1e6 / B.HBM_PEAK_GBPS}
# one epoch against the sequential result of the same triplets (lr 0.05: large steps, so that a lost update shows)
t.upload(P0, Q0)
s.epoch_throughput_async(0.05, 0.001, 0.001, chunk=chunk, groups=groups)
Pg, Qg = t.download()
Pr, Qr = P0.astype(np.float64), Q0.astype(np.float64)
O.bpr_sgd(Pr, Qr, u, ind, j, 0.05, 0.001, 0.001)
rel = lambda a, b, a0: float(np.linalg.norm(a - b) / np.linalg.norm(b - a0))
out["one_epoch_update_deviation_vs_sequential"] = {"P": rel(Pg, Pr, P0), "Q": rel(Qg, Qr, Q0)}       # relative to the MOVEMENT of the epoch
print(json.dumps(out), flush=True)
r = B.hbm_resident_roofline(capi, schedule="item")
out["hbm_resident"] = {k: r[k] for k in ("avg_launch_ms", "frac", "triplet_updates_per_s")}
print(json.dumps(out))
