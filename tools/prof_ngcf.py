#!/usr/bin/env python3
"""NGCF step (config #5) alone, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import NGCFTrainer, joint_norm_adjacency
from qrec_amd.synth import make_dataset
capi.init(0); rng = np.random.default_rng(0)
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
lim = np.sqrt(6 / 128); W = [[rng.uniform(-lim, lim, (64, 64)).astype(np.float32) for _ in range(2)] for _ in range(2)]
tr = NGCFTrainer((rng.standard_normal((nu, 64)) * 0.005).astype(np.float32), (rng.standard_normal((ni, 64)) * 0.005).astype(np.float32), W, adj, 0.002, 1e-3)
nn = d["train_u"].size; perm = rng.permutation(nn); B = 2048
du, di, dj = DB.from_numpy(d["train_u"][perm].astype(np.int32)), DB.from_numpy(d["train_i"][perm].astype(np.int32)), DB.from_numpy(rng.integers(0, ni, nn).astype(np.int32))
for k in range(5): tr.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, dj.ptr + 4 * k * B, B)
capi.device_sync(); t0 = time.perf_counter()
for k in range(60): tr.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, dj.ptr + 4 * k * B, B)
capi.device_sync(); print("ms_per_step", (time.perf_counter() - t0) / 60 * 1e3)
