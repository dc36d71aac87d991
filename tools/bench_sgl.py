#!/usr/bin/env python3
"""SGL step timing at the Yelp2018 shape with config/SGL.conf's hyper-parameters (L=3, edge dropout 0.1,
lambda 0.1, temp 0.2, d=64, batch 2048) + the per-epoch cost of the two sub-graphs both ways: exact mode (CPython replay + scipy arithmetic + SpMM
plan on the host) and throughput mode (round 6: drawn on the device as value arrays over the full graph's plan, graph.SubgraphSampler)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import SGLTrainer, SubgraphSampler, joint_norm_adjacency, sample_subgraph_edges, unique_first_appearance
from qrec_amd.synth import make_dataset
capi.init(0); d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; rng = np.random.default_rng(0)
uid, iid = d["train_u"].astype(np.int32), d["train_i"].astype(np.int32)
adj = joint_norm_adjacency(nu, ni, uid, iid)
tr = SGLTrainer((rng.standard_normal((nu, 64)) * 0.005).astype(np.float32), (rng.standard_normal((ni, 64)) * 0.005).astype(np.float32), adj, 3, 0.001, 1e-3, 0.1, 0.2, max_unique=4096)
import random; random.seed(0); st = capi.state_from_python(random.getstate())
t0 = time.perf_counter()
subs = []
for v in range(2):
    ku, ki = sample_subgraph_edges(st, uid, iid, nu, ni, 1, 0.1); subs.append(joint_norm_adjacency(nu, ni, ku, ki))
t_build = time.perf_counter() - t0
t0 = time.perf_counter(); tr.set_subgraphs(*subs); t_plan = time.perf_counter() - t0
n = uid.size; perm = rng.permutation(n); B = 2048; u, i = uid[perm], iid[perm]; j = rng.integers(0, ni, n).astype(np.int32)
du, di, dj = DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j); steps = 50
rows = [np.concatenate([unique_first_appearance(u[k * B:(k + 1) * B]), unique_first_appearance(i[k * B:(k + 1) * B]) + nu]).astype(np.int32) for k in range(steps)]
dr = [DB.from_numpy(r) for r in rows]
def step(k): tr.train_step_async(du.ptr + 4 * k * B, di.ptr + 4 * k * B, dj.ptr + 4 * k * B, B, dr[k], rows[k].size)
for k in range(5): step(k)
capi.device_sync(); t0 = time.perf_counter()
for k in range(steps): step(k)
capi.device_sync(); dt = (time.perf_counter() - t0) / steps
host_step_ms = dt * 1e3
# throughput mode: the same two sub-graphs on the device, the steps over the full graph's plan with their value arrays
t0 = time.perf_counter(); smp = SubgraphSampler(nu, ni, uid, iid, adj); capi.device_sync(); t_setup = time.perf_counter() - t0
vals = [smp.draw(1, 0.1, 7, (1 << 32) + k) for k in range(2)]; capi.device_sync()
e0, e1 = capi.Event(), capi.Event(); ts = []
for rep in range(12):
    e0.record(); vals = [smp.draw(1, 0.1, 7, (1 << 32) + 2 * rep + k, out=vals[k]) for k in range(2)]; e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
draw_ms = float(np.median(ts[2:]))
t0 = time.perf_counter(); tr.set_subgraph_values(vals[0], vals[1]); t_set = time.perf_counter() - t0
for k in range(5): step(k)
capi.device_sync(); t0 = time.perf_counter()
for k in range(steps): step(k)
capi.device_sync(); dt2 = (time.perf_counter() - t0) / steps
n_steps = -(-n // B)
print(json.dumps(dict(workload="SGL L=3 edge-dropout 0.1 d=64 batch=2048 Yelp2018 shape", ms_per_step=host_step_ms, triplets_per_s=B / dt, epoch_s=dt * n_steps,
                      exact_mode_host_subgraph_draw_and_build_s=t_build, exact_mode_host_plan_upload_s=t_plan,
                      exact_mode_host_share_of_epoch=(t_build + t_plan) / (dt * n_steps + t_build + t_plan),
                      throughput_mode=dict(device_draw_two_subgraphs_ms=draw_ms, host_set_values_ms=t_set * 1e3, one_time_setup_s=t_setup,
                                           ms_per_step_over_the_full_structure=dt2 * 1e3, epoch_s=dt2 * n_steps + draw_ms * 1e-3,
                                           per_epoch_augmentation_share=(draw_ms * 1e-3 + t_set) / (dt2 * n_steps + draw_ms * 1e-3 + t_set)),
                      contrast_rows=int(np.mean([r.size for r in rows])), losses=tr.losses())))
