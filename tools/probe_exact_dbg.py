#!/usr/bin/env python3
"""Where a step of the four-triplets-per-wavefront exact kernel spends its time: the kernel with one piece removed at a time
(QREC_EXACT_DBG bit mask, bpr_exact.hip; results of those runs are wrong by construction, only their time is read)."""
import json, os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
d = make_dataset("yelp2018"); U, I = d["n_users"], d["n_items"]
indptr, items = to_csr(U, d["train_u"], d["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32); n = items.size
j = capi.mt_bpr_sample_epoch(capi.state_from_python(random.Random(1).getstate()), indptr, items, I)
rng = np.random.default_rng(0); P0 = rng.random((U, 64)) / 3; Q0 = rng.random((I, 64)) / 3
NAMES = {0: "as shipped", 1: "no table stores", 2: "no table loads", 3: "no table loads, no stores", 4: "no exp / division", 8: "no barrier",
         7: "no table traffic, no exp"}
t = DeviceTables(P0, Q0, np.float64); s = BprSgd(t, u, items); s.set_negatives(j)
out = {}
with np.errstate(all="ignore"):
    for width in (4, 8):
        prep = None
        for dbg in (0, 1, 2, 3, 4, 8, 7):
            if dbg & 8 and width > 4:
                continue                      # without the barrier only a single wavefront is meaningful
            os.environ["QREC_EXACT_DBG"] = str(dbg)
            t.upload(P0, Q0)
            prep = s.prepare_ordered(j, width=width, slot=0, reorder=False)
            capi.device_sync()
            t0 = time.perf_counter(); s.run_prepared(prep, 0.01, 0.001, 0.001); capi.device_sync(); dt = time.perf_counter() - t0
            out[f"w{width}_dbg{dbg}"] = {"what": NAMES[dbg], "us_per_step": dt / prep["steps"] * 1e6, "steps": prep["steps"]}
            print(f"width {width}  dbg {dbg:2d}  {dt / prep['steps'] * 1e6:.3f} us/step   {NAMES[dbg]}", flush=True)
print(json.dumps(out))
