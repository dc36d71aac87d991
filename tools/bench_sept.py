#!/usr/bin/env python3
"""SEPT step time at the Yelp2018 shape (config/SEPT.conf options: 2 layers, ss_rate 0.005, drop 0.3, ins_cnt 10,
batch 2000, d = 50) with a synthetic follow graph (10 followees per user, uniform): recommendation-only steps, joint
steps, and the host cost of the per-epoch perturbed graph."""
import json, os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import SEPTTrainer, joint_norm_adjacency, sept_perturbed_adjacency, sept_user_views, unique_first_appearance
from qrec_amd.synth import make_dataset
capi.init(0); rng = np.random.default_rng(0); out = {}
d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]; n = nu + ni
uid, iid = d["train_u"].astype(np.int64), d["train_i"].astype(np.int64); nn = uid.size
fo = np.repeat(np.arange(nu), 10); fe = rng.integers(0, nu, fo.size)
t0 = time.perf_counter(); friend, sharing = sept_user_views(nu, ni, uid, iid, fo, fe); out["host_user_views_s"] = time.perf_counter() - t0
ip, ix, v = joint_norm_adjacency(nu, ni, uid, iid); adj = sp.csr_matrix((v, ix, ip), shape=(n, n))
words = capi.state_from_python((3, tuple([2 ** 31] + [0] * 623 + [624]), None))
t0 = time.perf_counter(); sub = sept_perturbed_adjacency(words, nu, ni, uid, iid, fo, fe, 0.3); out["host_perturbed_graph_s"] = time.perf_counter() - t0
dim, B, k = 50, 2000, 10
tr = SEPTTrainer((rng.standard_normal((nu, dim)) * 0.005).astype(np.float32), (rng.standard_normal((ni, dim)) * 0.005).astype(np.float32),
                 adj, friend, sharing, 2, 0.001, 0.001, 0.005, k, max_unique=B)
t0 = time.perf_counter(); tr.set_perturbed_graph(sub); out["host_plan_upload_s"] = time.perf_counter() - t0
out["nnz"] = dict(adj=int(adj.nnz), friend=int(friend.nnz), sharing=int(sharing.nnz), perturbed=int(sub.nnz))
perm = rng.permutation(nn)
u = uid[perm].astype(np.int32); i = iid[perm].astype(np.int32); j = rng.integers(0, ni, nn).astype(np.int32)
du, di, dj = DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j)
steps = 40
uu = [unique_first_appearance(u[s * B:(s + 1) * B]).astype(np.int32) for s in range(steps + 5)]
off = np.concatenate([[0], np.cumsum([x.size for x in uu])]); duu = DB.from_numpy(np.concatenate(uu))
for joint in (False, True):
    def step(s):
        tr.train_step_async(du.ptr + 4 * s * B, di.ptr + 4 * s * B, dj.ptr + 4 * s * B, B, joint, duu.ptr + 4 * int(off[s]), int(off[s + 1] - off[s]))
    for s in range(5): step(s)
    capi.device_sync(); t0 = time.perf_counter()
    for s in range(5, 5 + steps): step(s)
    capi.device_sync(); dt = (time.perf_counter() - t0) / steps
    rec, con = tr.losses()
    out["joint_step" if joint else "rec_step"] = dict(ms_per_step=dt * 1e3, triplets_per_s=B / dt, epoch_s=dt * -(-nn // B), rec_loss=rec, con_loss=con)
print(json.dumps(out))
