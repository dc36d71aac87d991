#!/usr/bin/env python3
"""Order-exact mode beyond one wavefront: epoch time of the scheduled kernel by width (wavefronts per step) and dtype at
the Yelp2018 shape, next to the one-wavefront walker; host cost of the schedule."""
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
out = {"n": int(n)}
# QREC_EXACT_KERNEL: reg = four triplets per wavefront (round 3, the default), w64 = one per wavefront (round 2)
combos = [(k, dt, w) for k in os.environ.get("PROBE_KERNELS", "reg,w64").split(",") for dt in ((np.float64, "f64"), (np.float32, "f32"))
          for w in ((4, 8, 12, 16) if k == "reg" else (8,))]
tabs = {}
for kernel, (dtype, tag0), width in combos:
    os.environ["QREC_EXACT_KERNEL"] = kernel
    tag = f"{kernel}_{tag0}"
    if tag0 not in tabs:
        tabs[tag0] = DeviceTables(P0, Q0, dtype)
    t = tabs[tag0]; s = BprSgd(t, u, items); s.set_negatives(j)
    for width in (width,):
        t.upload(P0, Q0); capi.device_sync()
        t0 = time.perf_counter(); ent, off = capi.bpr_exact_schedule(s.h_u, s.h_i, s.h_j, U, I, width) if width > 1 else (None, None); ts = time.perf_counter() - t0
        t0 = time.perf_counter(); loss = s.epoch_ordered(0.01, 0.001, 0.001, width=width); dt = time.perf_counter() - t0
        steps = getattr(s, "exact_steps", n) if width > 1 else n
        out[f"{tag}_w{width}"] = {"epoch_s_incl_host": dt, "host_schedule_s": ts, "device_s": dt - ts, "steps": int(steps),
                                  "us_per_step": (dt - ts) / steps * 1e6, "M_triplets_per_s_device": n / (dt - ts) / 1e6,
                                  "M_triplets_per_s_incl_host": n / dt / 1e6, "loss": loss}
        print(tag, width, out[f"{tag}_w{width}"], flush=True)
print(json.dumps(out))
