#!/usr/bin/env python3
"""How launch-bound is a LightGCN step on a graph of the reference's own dataset sizes (lastfm / FilmTrust shape)?
wall time per step (host enqueue rate vs device) against the sum of kernel durations (rocprofv3 --kernel-trace)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import LightGCNTrainer, joint_norm_adjacency
capi.init(0); rng = np.random.default_rng(0); out = {}
for name, nu, ni, E, d, B in (("lastfm", 1892, 17632, 92834, 50, 2000), ("filmtrust", 1508, 2071, 35497, 50, 2000)):
    uid = rng.integers(0, nu, E); iid = rng.integers(0, ni, E)
    tr = LightGCNTrainer((rng.standard_normal((nu, d)) * 0.005).astype(np.float32), (rng.standard_normal((ni, d)) * 0.005).astype(np.float32),
                         joint_norm_adjacency(nu, ni, uid, iid), 2, 0.001, 0.001)
    du, di, dj = DB.from_numpy(uid.astype(np.int32)), DB.from_numpy(iid.astype(np.int32)), DB.from_numpy(rng.integers(0, ni, E).astype(np.int32))
    nb = E // B
    for s in range(10): tr.train_step_async(du.ptr + 4 * (s % nb) * B, di.ptr + 4 * (s % nb) * B, dj.ptr + 4 * (s % nb) * B, B)
    capi.device_sync(); t0 = time.perf_counter(); steps = 400
    for s in range(steps): tr.train_step_async(du.ptr + 4 * (s % nb) * B, di.ptr + 4 * (s % nb) * B, dj.ptr + 4 * (s % nb) * B, B)
    t_enq = time.perf_counter() - t0
    capi.device_sync(); dt = (time.perf_counter() - t0) / steps
    out[name] = dict(ms_per_step=dt * 1e3, host_enqueue_ms_per_step=t_enq / steps * 1e3, steps_per_epoch=-(-E // B))
print(json.dumps(out))
