"""world_size-2 test of the multi-GPU scheme on CPU (gloo): user sharding is a partition,
and a sharded step with the replicated-Q delta all-reduce equals the single-process
definition  Q_start + sum_r (Q_r - Q_start)  with each shard's P rows updated locally."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c as O
from qrec_amd.dist import ReplicatedTableSync, shard_positive_csr, user_block
from qrec_amd.synth import make_dataset, to_csr


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    d = make_dataset("tiny")
    indptr, ind = to_csr(d["n_users"], d["train_u"], d["train_i"])
    rng = np.random.default_rng(0)
    P0 = rng.random((d["n_users"], 16)) / 3; Q0 = rng.random((d["n_items"], 16)) / 3
    return d, indptr, ind, P0, Q0


def _local_epoch(indptr, ind, lo, hi, P0, Q, n_items, seed):
    """one rank's work: its users' triplets, own sampler stream, order-exact on its replica"""
    lp, li = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64), np.ascontiguousarray(ind[indptr[lo]:indptr[hi]])
    u = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(lp)).astype(np.int32)
    j = O.bpr_sample_epoch(O.MT.cpython_seed(seed), lp, li, n_items)
    P = P0[lo:hi].copy()
    O.bpr_sgd(P, Q, u, li, j, 0.05, 0.01, 0.01)
    return P


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, indptr, ind, P0, Q0 = _problem()
    lo, hi, lp, li = shard_positive_csr(indptr, ind, world, rank)
    q = torch.from_numpy(Q0.copy())
    sync = ReplicatedTableSync(q)
    P = None
    for step in range(2):
        P0_step = P0 if P is None else np.concatenate([P0[:lo], P, P0[hi:]])
        P = _local_epoch(indptr, ind, lo, hi, P0_step, q.numpy(), d["n_items"], 100 * step + rank)
        sync.sync()
    out[rank] = (lo, hi, P, q.numpy().copy())
    dist.barrier(); dist.destroy_process_group()


def test_user_blocks_partition():
    for n in (0, 1, 7, 31668, 10_000_000):
        for w in (1, 2, 3, 8):
            blocks = [user_block(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    d, indptr, ind, _, _ = _problem()
    parts = [shard_positive_csr(indptr, ind, 3, r) for r in range(3)]
    assert np.array_equal(np.concatenate([p[3] for p in parts]), ind)
    assert sum(p[2][-1] for p in parts) == ind.size


def test_two_rank_step_equals_definition():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d, indptr, ind, P0, Q0 = _problem()
    # single-process statement of the same semantics
    Q = Q0.copy(); P = P0.copy()
    for step in range(2):
        deltas = []
        newP = P.copy()
        for r in range(world):
            lo, hi = user_block(d["n_users"], world, r)
            Qr = Q.copy()
            newP[lo:hi] = _local_epoch(indptr, ind, lo, hi, P, Qr, d["n_items"], 100 * step + r)
            deltas.append(Qr - Q)
        Q = Q + sum(deltas); P = newP
    for r in range(world):
        lo, hi, Pr, Qr = out[r]
        np.testing.assert_allclose(Qr, Q, rtol=1e-12, atol=1e-15)       # replicas agree and equal the definition
        np.testing.assert_allclose(Pr, P[lo:hi], rtol=1e-12, atol=1e-15)
    assert not np.allclose(Q, Q0)


# ---- graph models: batch-sharded data parallelism (qrec_amd.dist.BatchParallel) -----------------------------------
def _graph_problem():
    import scipy.sparse as sp
    from oracle import tfmodels as T
    from qrec_amd.graph import joint_norm_adjacency
    d = make_dataset("tiny")
    indptr, idx, val = joint_norm_adjacency(d["n_users"], d["n_items"], d["train_u"], d["train_i"])
    n = d["n_users"] + d["n_items"]
    A = sp.csr_matrix((val, idx, indptr), shape=(n, n))
    rng = np.random.default_rng(5)
    U0 = (rng.standard_normal((d["n_users"], 8)) * 0.1).astype(np.float32)
    V0 = (rng.standard_normal((d["n_items"], 8)) * 0.1).astype(np.float32)
    B = 96
    u = rng.integers(0, d["n_users"], B); i = rng.integers(0, d["n_items"], B); j = rng.integers(0, d["n_items"], B)
    return T.LightGCN(U0, V0, A, 2, lr=0.01, reg=1e-3), u, i, j


def _graph_worker(rank, world, port, out):
    from qrec_amd.dist import BatchParallel, is_output_rank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = BatchParallel(device_index=0)
    model, u, i, j = _graph_problem()
    steps = []
    for step in range(2):
        lo, cnt = dp.share(u.size)
        loss, g = model.loss_and_grad(u[lo:lo + cnt], i[lo:lo + cnt], j[lo:lo + cnt])     # this rank's share of the step
        g = dp.all_reduce_host(g); loss = float(dp.all_reduce_host(np.array([loss], np.float64))[0])
        model.opt.step(model.E, g)                                                          # same update on every replica
        steps.append(loss)
    out[rank] = (dp.share(u.size), steps, model.E.copy(), is_output_rank())
    dist.barrier(); dist.destroy_process_group()


def test_graph_batch_parallel_equals_the_whole_step():
    """two ranks, each with a share of the step's rows + gradient all-reduce == one process on the whole step"""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_graph_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    model, u, i, j = _graph_problem()
    want = [model.train_step(u, i, j) for _ in range(2)]
    shares = [out[r][0] for r in range(world)]
    assert shares[0][0] == 0 and shares[0][1] + shares[1][1] == u.size and shares[1][0] == shares[0][1]
    assert np.array_equal(out[0][2], out[1][2])                         # replicas identical
    np.testing.assert_allclose(out[0][1], want, rtol=1e-5)
    np.testing.assert_allclose(out[0][2], model.E, rtol=0, atol=2e-5)
    assert out[0][3] is True and out[1][3] is False                     # rank 0 alone writes result files
