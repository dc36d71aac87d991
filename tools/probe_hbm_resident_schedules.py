#!/usr/bin/env python3
"""bench.py's roofline_hbm_resident workload (config #4 single-GPU slice: 1.25 M x 1 M, d = 128, 25 M triplets, 1.15 GB of tables)
under every throughput schedule: user-major (what the bench line reports), item-major, item-deferred."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from qrec_amd import capi
capi.init(0)
out = {}
for sched in ("user", "item", "item-deferred"):
    r = B.hbm_resident_roofline(capi, schedule=sched)
    out[sched] = {k: r[k] for k in ("avg_launch_ms", "frac", "achieved", "triplet_updates_per_s")}
    print(sched, out[sched], flush=True)
print(json.dumps(out))
