#!/usr/bin/env python3
"""Round 6: where does P[u]-by-load+store stop paying?  The item-major kernel at d = 128 with 20 triplets per user, 1 M items, by number of users
(= size of the user table the triplets scatter over): epoch time with atomic deltas and with load + store.  One JSON line per size."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B                                              # noqa: E402
from qrec_amd import capi                                      # noqa: E402
from qrec_amd.engine import BprSgd, DeviceTables               # noqa: E402

capi.init(0)
I, d, per = 1_000_000, 128, 20
sizes = [int(x) for x in sys.argv[1:]] or [1_250_000, 2_500_000, 5_000_000, 10_000_000]
blk = (np.random.default_rng(1).random((100_000, d)) / 3).astype(np.float32)
for U in sizes:
    rng = np.random.default_rng(U)
    n = U * per
    u = np.repeat(np.arange(U, dtype=np.int32), per); i = rng.integers(0, I, n, dtype=np.int32); j = rng.integers(0, I, n, dtype=np.int32)
    P = np.empty((U, d), np.float32); Q = np.empty((I, d), np.float32)
    for a in (P, Q):
        for k in range(0, a.shape[0], 100_000):
            a[k:k + 100_000] = blk[:min(100_000, a.shape[0] - k)]
    t = DeviceTables(P, Q, np.float32); del P, Q
    row = {"users": U, "triplets": n, "P_table_GB": U * d * 4 / 1e9}
    for sched, pol in (("item", "atomic"), ("item", "rmw"), ("user", "atomic")):
        s = BprSgd(t, u, i, None, schedule=sched, p_update=pol); s.set_negatives(j)
        ms = B._time_events(capi, lambda: s.epoch_throughput_async(0.01, 0.001, 0.001), 3, warm=1)
        row[f"{sched}/{pol}"] = {"ms": ms, "frac_of_8TBps": n * B.bytes_per_triplet(d) / ms / 1e6 / 8000.0}
        del s
    print(json.dumps(row), flush=True)
    del t
