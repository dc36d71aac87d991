#!/usr/bin/env python3
"""bench.py's roofline_hbm_resident workload (config #4 single-GPU slice: 1.25 M x 1 M, d = 128, 25 M triplets, 1.15 GB of tables)
under every throughput schedule: user-major, item-major with atomic P[u] updates, item-major with P[u] by load + store (round 6)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from qrec_amd import capi
capi.init(0)
out = {}
for sched, p_update in (("user", "atomic"), ("item", "atomic"), ("item", "rmw")):
    r = B.hbm_resident_roofline(capi, schedule=sched, p_update=p_update)
    out[f"{sched}/{p_update}"] = {k: r[k] for k in ("avg_launch_ms", "frac", "achieved", "triplet_updates_per_s")}
    print(sched, p_update, out[f"{sched}/{p_update}"], flush=True)
print(json.dumps(out))
