#!/usr/bin/env python3
"""One order-exact epoch (300 k triplets, Yelp-sized fp64 tables) for counter collection."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.engine import BprSgd, DeviceTables
capi.init(0); rng = np.random.default_rng(0)
U, I, n = 31668, 38048, 300_000
u = np.sort(rng.integers(0, U, n)).astype(np.int32); i = rng.integers(0, I, n).astype(np.int32); j = rng.integers(0, I, n).astype(np.int32)
t = DeviceTables(rng.random((U, 64)) / 3, rng.random((I, 64)) / 3, np.float64); s = BprSgd(t, u, i); s.set_negatives(j)
t0 = time.perf_counter(); s.epoch_ordered(0.01, 0.001, 0.001); capi.device_sync(); print("ns_per_triplet", (time.perf_counter() - t0) / n * 1e9)
