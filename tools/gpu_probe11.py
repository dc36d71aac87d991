#!/usr/bin/env python3
"""Where does the step time go?  SGD kernel time (events) and whole-step time for: sampler overlapped on the
side stream / serial on the main stream / absent, x epoch close on the device / on the host."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
from qrec_amd.interactions import CSR
from qrec_amd.synth import make_dataset, to_csr
capi.init(0)
data = make_dataset("yelp2018"); U, I = data["n_users"], data["n_items"]
indptr, items = to_csr(U, data["train_u"], data["train_i"])
u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32); n = items.size
rng = np.random.default_rng(0)
P0 = (rng.random((U, 64)) / 3).astype(np.float32); Q0 = (rng.random((I, 64)) / 3).astype(np.float32)
out = {}
for sampler in ("side", "main", "none"):
    for close in ("device", "host"):
        t = DeviceTables(P0, Q0, np.float32); sgd = BprSgd(t, u, items, CSR(indptr, items), schedule="item")
        sgd.sample_negatives_device(1, 0); sgd.start_device_driver(0.01, 64)
        W, K = 3, 20
        ev = [(capi.Event(), capi.Event()) for _ in range(W + K)]
        if sampler == "side": sgd.prefetch_negatives_device(1, 0)
        lr = 0.01
        def step(k):
            global lr
            if sampler == "side": sgd.take_prefetched_negatives(k)
            elif sampler == "main": sgd.sample_negatives_device(1, k)
            if close == "device":
                sgd.epoch_device_async(0.001, 0.001, 1.0, tol=0.0, events=ev[k])
            else:
                capi._check(capi.load().qrec_memset(sgd.d_stats.ptr, 0, 8, None))
                ev[k][0].record(); sgd.epoch_throughput_async(lr, 0.001, 0.001); ev[k][1].record(); sgd.mark_negatives_consumed()
            if sampler == "side": sgd.prefetch_negatives_device(1, k + 1)
            if close == "host":
                sgd.epoch_stats()
        for k in range(W): step(k)
        capi.device_sync(); t0 = time.perf_counter()
        for k in range(W, W + K): step(k)
        capi.device_sync(); dt = (time.perf_counter() - t0) / K
        out[f"sampler={sampler},close={close}"] = dict(step_ms=round(dt * 1e3, 3), sgd_kernel_ms=round(float(np.mean([ev[k][1].elapsed_ms_since(ev[k][0]) for k in range(W, W + K)])), 3))
print(json.dumps(out, indent=0))
