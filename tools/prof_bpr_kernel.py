#!/usr/bin/env python3
"""A few launches of the item-major BPR kernel for a profiler to look at (rocprofv3 --pmc passes, tools/gpu_steps.sh step bpr-counters):
    prof_bpr_kernel.py <yelp|hbm> <atomic|rmw> [epochs]
yelp = the bench line's workload (Yelp2018 shape, d = 64, tables cache resident); hbm = the HBM-resident slice of config #4 (1.25 M x 1 M,
d = 128, 25 M triplets).  Prints the HIP-event time per launch, so that a counter pass carries its own duration."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                              # noqa: E402
from qrec_amd import capi                                      # noqa: E402
from qrec_amd.engine import BprSgd, DeviceTables               # noqa: E402
from qrec_amd.interactions import CSR                          # noqa: E402
from qrec_amd.synth import make_dataset, to_csr                # noqa: E402

which, pol = sys.argv[1], sys.argv[2]
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
capi.init(0)
rng = np.random.default_rng(0)
if which == "yelp":
    d = make_dataset("yelp2018"); U, I, dim = d["n_users"], d["n_items"], 64
    indptr, ind = to_csr(U, d["train_u"], d["train_i"])
    u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
    t = DeviceTables((rng.random((U, dim)) / 3).astype(np.float32), (rng.random((I, dim)) / 3).astype(np.float32), np.float32)
    s = BprSgd(t, u, ind, CSR(indptr, ind), schedule="item", p_update=pol)
    s.sample_negatives_device(7, 0)
else:
    U, I, n, dim = 1_250_000, 1_000_000, 25_000_000, 128
    u = np.sort(rng.integers(0, U, n, dtype=np.int32)); ind = rng.integers(0, I, n, dtype=np.int32)
    blk = (rng.random((50_000, dim)) / 3).astype(np.float32)
    P = np.empty((U, dim), np.float32); Q = np.empty((I, dim), np.float32)
    for a in (P, Q):
        for k in range(0, a.shape[0], 50_000):
            a[k:k + 50_000] = blk[:min(50_000, a.shape[0] - k)]
    t = DeviceTables(P, Q, np.float32); s = BprSgd(t, u, ind, None, schedule="item", p_update=pol)
    s.set_negatives(rng.integers(0, I, n, dtype=np.int32))
chunk, groups = s.launch_grid(None)
ms = B._time_events(capi, lambda: s.epoch_throughput_async(0.01, 0.001, 0.001, chunk=chunk, groups=groups), epochs, warm=2)
n = int(u.size)
print(json.dumps({"workload": which, "p_update": s.p_update, "collision_density": s.collision, "triplets": n, "ms_per_launch": ms,
                  "algorithmic_bytes": n * B.bytes_per_triplet(dim), "roofline_frac": n * B.bytes_per_triplet(dim) / ms / 1e6 / B.HBM_PEAK_GBPS}))
