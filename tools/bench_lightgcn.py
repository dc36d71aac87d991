#!/usr/bin/env python3
"""LightGCN training-step measurement at BASELINE.json config #3 (Yelp2018 shape, L=3, d=64,
batch 2048).  Prints one JSON object; development/measurement tool (the contract bench is
bench.py).  --cpu also times the numpy/scipy restatement for a few steps."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import LightGCNTrainer, joint_norm_adjacency
from qrec_amd.synth import make_dataset
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--batch", type=int, default=2048); ap.add_argument("--cpu", action="store_true"); ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--shape", default="yelp2018", help="yelp2018 (structureless Zipf) or yelp2018-clustered (64 planted communities)")
a = ap.parse_args()
capi.init(0)
d = make_dataset(a.shape); nu, ni = d["n_users"], d["n_items"]
adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
rng = np.random.default_rng(0)
U0 = (rng.standard_normal((nu, a.dim)) * 0.005).astype(np.float32); V0 = (rng.standard_normal((ni, a.dim)) * 0.005).astype(np.float32)
tr = LightGCNTrainer(U0, V0, adj, a.layers, lr=0.001, reg=1e-4)
n = d["train_u"].size
perm = rng.permutation(n)
u = DB.from_numpy(d["train_u"][perm].astype(np.int32)); i = DB.from_numpy(d["train_i"][perm].astype(np.int32)); j = DB.from_numpy(rng.integers(0, ni, n).astype(np.int32))
B = a.batch
def step(k):
    s = (k * B) % (n - B)
    tr.train_step_async(u.ptr + 4 * s, i.ptr + 4 * s, j.ptr + 4 * s, B)
for k in range(5): step(k)
capi.device_sync()
# whole step
t0 = time.perf_counter()
for k in range(a.steps): step(k)
t_host = (time.perf_counter() - t0) / a.steps
capi.device_sync(); dt = (time.perf_counter() - t0) / a.steps
# with per-step loss read-back (what the reference's print forces)
t0 = time.perf_counter()
for k in range(a.steps): step(k); tr.loss()
dt_sync = (time.perf_counter() - t0) / a.steps
# SpMM alone
e0, e1 = capi.Event(), capi.Event(); ts = []
for rep in range(20):
    e0.record(); capi.spmm_csr(tr.plan, tr.E, tr.A, tr.ld, d_accum=tr.S); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
spmm_ms = float(np.median(ts)); nnz = tr.plan.nnz; N = nu + ni
ts = []
for rep in range(20):
    e0.record(); capi.adam_step(tr.E, tr.m, tr.v, tr.A, N * tr.ld, 0.25, 1e-9); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
adam_ms = float(np.median(ts))
alg = tr.plan.bytes_algorithmic(a.dim) + N * a.dim * 4   # + accum RMW read
out = dict(workload=f"LightGCN L={a.layers} d={a.dim} batch={B} {a.shape}-shape N={N} nnz={nnz}", spmm_chunks=tr.plan.chunks, ms_per_step=dt * 1e3, host_enqueue_ms_per_step=t_host * 1e3, ms_per_step_with_loss_readback=dt_sync * 1e3,
           triplets_per_s=B / dt, steps_per_epoch=-(-n // B), epoch_s=dt * -(-n // B),
           spmm_ms=spmm_ms, spmm_algorithmic_GBps=alg / spmm_ms / 1e6, spmm_gather_GBps=(nnz * (8 + a.dim * 4) + 2 * N * a.dim * 4) / spmm_ms / 1e6,
           spmm_gflops=2 * nnz * a.dim / spmm_ms / 1e6, adam_ms=adam_ms, adam_GBps=7 * 4 * N * tr.ld / adam_ms / 1e6, segments=tr.plan.n_segs, long_rows=tr.plan.n_long)
if a.cpu:
    import scipy.sparse as sp
    from oracle import tfmodels as T
    A = sp.csr_matrix((adj[2], adj[1], adj[0]), shape=(N, N))
    ref = T.LightGCN(U0, V0, A, a.layers, 0.001, 1e-4)
    uu, ii = d["train_u"][perm][:B].astype(np.int32), d["train_i"][perm][:B].astype(np.int32); jj = rng.integers(0, ni, B)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 10: ref.train_step(uu, ii, jj); k += 1
    out["cpu_restatement_ms_per_step"] = (time.perf_counter() - t0) / k * 1e3; out["cpu_steps_timed"] = k
    out["speedup_vs_cpu_restatement"] = out["cpu_restatement_ms_per_step"] / out["ms_per_step"]
print(json.dumps(out))
