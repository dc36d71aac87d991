"""GPU tests of the multi-GPU BPR path on the one device the test boxes have:

* the kernels around the collectives (delta / apply, the shard plan, row gather, return of row deltas) against
  tests/hostkern.py, the numpy statement of the same entry points;
* the real RCCL binding (qrec_comm_*) with a world of one;
* the row-sharded exchange with G logical ranks IN ONE PROCESS (threads + an in-process fake collective that copies
  device to device), the real kernels and the real SGD kernel, against the single-process definition
      per batch:  Q <- Q + sum_r (cache_r_after - cache_r_before);
* bench.py's N > 1 paths with two real processes on the one device (staged gloo transport) and with
  QREC_FORCE_DIST=1 (real RCCL, world 1)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import c as O
from qrec_amd import capi
from qrec_amd import dist as qd
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.dist import user_block
from qrec_amd.synth import make_dataset, to_csr
from tests import hostkern as HK

from helpers import check, check_rel, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def test_table_delta_and_apply():
    rng = np.random.default_rng(0)
    n = 4 * 12345
    table, start, other = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    d_t, d_s, d_d = DB.from_numpy(table), DB.from_numpy(start), DB(n, np.float32)
    capi.table_delta(d_t, d_s, d_d, n)
    assert np.array_equal(d_d.numpy(), table - start)
    d_d.upload(other)
    capi.table_apply(d_t, d_s, d_d, n)
    assert np.array_equal(d_s.numpy(), start + other) and np.array_equal(d_t.numpy(), start + other)
    with pytest.raises(capi.QRecError):
        capi.table_delta(d_t, d_s, d_d, 7)


@pytest.mark.parametrize("ld", [32, 64, 256])
def test_row_subset_reconciliation_and_batch_row_kernels(ld):
    """qrec_batch_rows_gather / _scatter_add -- the rows {u, nu + i, nu + j} of a batch out of / into a block [lo, hi) of a row-partitioned
    table, repeated rows adding up."""
    rng = np.random.default_rng(ld)
    n_rows = 5000
    table = rng.standard_normal((n_rows, ld)).astype(np.float32)
    # (b)
    nu, B, lo, hi = 2000, 333, 1500, 3700
    u = rng.integers(0, nu, B).astype(np.int32); i = rng.integers(0, n_rows - nu, B).astype(np.int32); j = rng.integers(0, n_rows - nu, B).astype(np.int32)
    ids = np.concatenate([u, nu + i, nu + j]).astype(np.int64)
    d_u, d_i, d_j = DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j)
    block = table[lo:hi].copy(); d_b = DB.from_numpy(block); d_o = DB((3 * B, ld), np.float32)
    capi.batch_rows_gather(d_b, ld, lo, hi, d_u, d_i, d_j, B, nu, d_o)
    own = (ids >= lo) & (ids < hi)
    want = np.where(own[:, None], table[np.clip(ids, 0, n_rows - 1)], 0).astype(np.float32)
    assert np.array_equal(d_o.numpy(), want) and 0 < own.sum() < ids.size
    src = rng.standard_normal((3 * B, ld)).astype(np.float32)
    capi.batch_rows_scatter_add(d_b, ld, lo, hi, d_u, d_i, d_j, B, nu, DB.from_numpy(src))
    ref = block.astype(np.float64)
    np.add.at(ref, ids[own] - lo, src[own].astype(np.float64))
    check("rel_err(batch_rows_scatter_add, numpy add.at)", rel_err(d_b.numpy(), ref), 1e-6)
    with pytest.raises(capi.QRecError):
        capi.batch_rows_gather(d_b, 48, lo, hi, d_u, d_i, d_j, B, nu, d_o)


@pytest.mark.parametrize("n_items,world,n", [(200, 1, 500), (200, 2, 500), (38048, 8, 150_000), (1_000_003, 8, 400_000),
                                              (5, 8, 40), (3000, 3, 0), (1025, 2, 1)])
def test_shard_plan_matches_the_host_statement(n_items, world, n):
    rng = np.random.default_rng(n_items + world)
    i = rng.integers(0, n_items, n).astype(np.int32); j = rng.integers(0, n_items, n).astype(np.int32)
    cap = max(min(2 * n, n_items), 1)
    d_scr = DB(capi.shard_plan_scratch_bytes(n_items, world), np.uint8)
    d_req, d_cnt = DB(cap, np.int32), DB(world, np.int32)
    d_ci, d_cj = DB(max(n, 1), np.int32), DB(max(n, 1), np.int32)
    d_i, d_j = DB.from_numpy(i if n else np.zeros(1, np.int32)), DB.from_numpy(j if n else np.zeros(1, np.int32))
    for _ in range(2):            # twice: the scratch map is reused from batch to batch
        capi.shard_plan_batch(d_i, d_j, n, n_items, world, d_scr, d_req, d_cnt, d_ci, d_cj)
    h = {k: HK.DeviceBuffer(s, np.int32) for k, s in (("req", cap), ("cnt", world), ("ci", max(n, 1)), ("cj", max(n, 1)))}
    HK.shard_plan_batch(HK.DeviceBuffer.from_numpy(i), HK.DeviceBuffer.from_numpy(j), n, n_items, world, None, h["req"], h["cnt"],
                        h["ci"], h["cj"])
    cnt = d_cnt.numpy()
    assert np.array_equal(cnt, h["cnt"].a)
    r = int(cnt.sum())
    assert np.array_equal(d_req.numpy()[:r], h["req"].a[:r])
    if n:
        assert np.array_equal(d_ci.numpy()[:n], h["ci"].a[:n]) and np.array_equal(d_cj.numpy()[:n], h["cj"].a[:n])
        # what the plan is for: slot -> (owner, row) -> item id gives the triplet's item back
        owner = np.repeat(np.arange(world), cnt)
        item_of_slot = d_req.numpy()[:r].astype(np.int64) * world + owner
        assert np.array_equal(item_of_slot[d_ci.numpy()[:n]], i) and np.array_equal(item_of_slot[d_cj.numpy()[:n]], j)


@pytest.mark.parametrize("ld", [32, 64, 128, 256])
def test_row_gather_and_return_of_deltas(ld):
    rng = np.random.default_rng(ld)
    rows_total, n = 5000, 3000
    table = rng.standard_normal((rows_total, ld)).astype(np.float32)
    rows = rng.integers(0, rows_total, n).astype(np.int32)             # with repeats: several ranks return the same row
    d_table, d_rows, d_out = DB.from_numpy(table), DB.from_numpy(rows), DB((n, ld), np.float32)
    capi.gather_rows(d_table, ld, d_rows, n, d_out)
    assert np.array_equal(d_out.numpy(), table[rows])
    fresh = table[rows] + (rng.standard_normal((n, ld)) * 1e-2).astype(np.float32)
    fresh[::7] = table[rows][::7]                                      # untouched rows come back as they left
    capi.scatter_add_row_deltas(d_table, ld, d_rows, n, DB.from_numpy(fresh), d_out)
    want = table.astype(np.float64); np.add.at(want, rows, (fresh - table[rows]).astype(np.float64))
    np.testing.assert_allclose(d_table.numpy(), want, rtol=0, atol=2e-6)
    untouched = np.setdiff1d(np.arange(rows_total), rows)
    assert np.array_equal(d_table.numpy()[untouched], table[untouched])
    capi.gather_rows(d_table, ld, d_rows, 0, d_out); capi.scatter_add_row_deltas(d_table, ld, d_rows, 0, d_out, d_out)
    with pytest.raises(capi.QRecError):
        capi.gather_rows(d_table, 48, d_rows, n, d_out)


def test_rccl_binding_with_a_world_of_one():
    """the real communicator (qrec_comm_init on librccl): every collective of the C ABI, degenerate but live"""
    path, version = capi.comm_library()
    assert "rccl" in path and version > 20000
    comm = capi.Comm(1, 0, capi.comm_unique_id(), identity_shortcut=False)
    rng = np.random.default_rng(1)
    a = rng.standard_normal(4096).astype(np.float32); s = np.array([1.5, -2.0, 7.0])
    d_a, d_s = DB.from_numpy(a), DB.from_numpy(s)
    comm.allreduce(d_a, a.size, capi.F32)
    comm.allreduce_pair(d_a, a.size, capi.F32, d_s, 2, capi.F64)
    assert np.array_equal(d_a.numpy(), a) and np.array_equal(d_s.numpy(), s)
    d_b = DB(a.size, np.float32)
    comm.allgather(d_a, d_b, a.size, capi.F32)
    assert np.array_equal(d_b.numpy(), a)
    d_b.fill_bytes(0); comm.reduce_scatter(d_a, d_b, a.size, capi.F32)
    assert np.array_equal(d_b.numpy(), a)
    rows = rng.integers(0, 1 << 30, (100, 64)).astype(np.int32)
    d_r, d_o = DB.from_numpy(rows), DB((100, 64), np.int32)
    st = capi.Stream()
    comm.alltoall_rows(d_r, [100], d_o, [100], 256, st); st.sync()
    assert np.array_equal(d_o.numpy(), rows)
    comm.alltoall_rows(d_r, [0], d_o, [0], 256)
    # arbitrary segments in one fused launch (the request ids of all batches of an epoch): two segments to itself, out of order
    d_o.fill_bytes(0)
    comm.sendrecv_segments(d_r, [(0, 256 * 10, 256 * 5), (0, 0, 256 * 3)], d_o, [(0, 0, 256 * 5), (0, 256 * 50, 256 * 3)], st); st.sync()
    got = d_o.numpy()
    assert np.array_equal(got[:5], rows[10:15]) and np.array_equal(got[50:53], rows[:3]) and not got[5:50].any()
    with pytest.raises(capi.QRecError):
        comm.sendrecv_segments(d_r, [(3, 0, 16)], d_o, [], st)                 # peer outside the world
    with pytest.raises(ValueError):
        comm.alltoall_rows(d_r, [1, 2], d_o, [1, 2], 256)
    # dist.preflight (round 5): the checked first contact bench.py --gpus N makes, here through RCCL itself (no identity short cut)
    from qrec_amd import dist as qd
    pre = qd.preflight(comm, stream=st, timeout_s=60)
    assert pre["ok"] and pre["world"] == 1 and pre["alltoall_rows_sent"] == 1 and pre["allreduce_floats"] == 256
    with pytest.raises(capi.QRecError):
        comm.allreduce(d_a, a.size, 9)
    comm.destroy()
    with pytest.raises(capi.QRecError):
        capi.Comm(2, 5, capi.comm_unique_id())


from tests.logical_ranks import ThreadComm, _Group      # G logical ranks in one process


def _tiny_problem(dim):
    d = make_dataset("small")
    indptr, ind = to_csr(d["n_users"], d["train_u"], d["train_i"])
    rng = np.random.default_rng(4)
    return d, indptr, ind, rng.random((d["n_users"], dim)) / 3, (rng.random((d["n_items"], dim)) / 3).astype(np.float32)


def _rank_triplets(indptr, ind, lo, hi, n_items, seed, schedule):
    lp, li = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64), np.ascontiguousarray(ind[indptr[lo]:indptr[hi]])
    u = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(lp)).astype(np.int32)
    j = O.bpr_sample_epoch(O.MT.cpython_seed(seed), lp, li, n_items)
    if schedule == "item":
        perm = np.argsort(li, kind="stable")
        u, li, j = u[perm], li[perm], j[perm]
    return np.ascontiguousarray(u), np.ascontiguousarray(li), np.ascontiguousarray(j)


@pytest.mark.parametrize("world,dim", [(2, 64), (3, 50), (1, 8)])
@pytest.mark.parametrize("exchange", ["allgather", "referenced"])
def test_row_partitioned_ngcf_step_equals_the_single_gpu_step(world, dim, exchange, monkeypatch):
    """RowPartitionedNGCFTrainer with G logical ranks: every rank holds its rows of E_0, the Adam slots, the adjacency and
    the per-layer tables; weights replicated, their gradients all-reduced.  Same batch, same injected dropout decisions,
    same step as NGCFTrainer on one GPU: losses, the rank's rows of E_0, the four weights and the inference embeddings
    agree to fp32 summation order (atomics of the batch gradient, the slab order of the weight gradients)."""
    monkeypatch.setenv("QREC_GRAPH_EXCHANGE", exchange)      # referenced: only the remote rows a rank's block refers to travel (round 3)
    from qrec_amd.graph import NGCFTrainer, RowPartitionedNGCFTrainer, joint_norm_adjacency
    from qrec_amd.engine import padded_ld
    d = make_dataset("small")
    nu, ni = d["n_users"], d["n_items"]
    N = nu + ni
    adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
    rng = np.random.default_rng(world + dim)
    U0 = (rng.standard_normal((nu, dim)) * 0.1).astype(np.float32); V0 = (rng.standard_normal((ni, dim)) * 0.1).astype(np.float32)
    lim = np.sqrt(6.0 / (2 * dim))
    W = [[rng.uniform(-lim, lim, (dim, dim)).astype(np.float32) for _ in range(2)] for _ in range(2)]
    B, ld = 512, padded_ld(dim, np.float32)
    steps = []
    for _ in range(2):
        sel = rng.integers(0, d["train_u"].size, B)
        steps.append((d["train_u"][sel].astype(np.int32), d["train_i"][sel].astype(np.int32), rng.integers(0, ni, B).astype(np.int32),
                      [np.pad((rng.random((N, dim)) < 0.9).astype(np.float32), ((0, 0), (0, ld - dim))) for _ in range(2)]))
    one = NGCFTrainer(U0, V0, W, adj, lr=0.002, reg=1e-3)
    losses_one = []
    for u, i, j, masks in steps:
        one.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B, masks=[DB.from_numpy(m) for m in masks])
        losses_one.append(one.loss())
    U1, V1, W1 = one.parameters()
    E_one = np.concatenate([U1, V1])
    Ui, Vi = one.inference_embeddings()
    group = _Group(world)
    result, errors = [None] * world, []

    def run(rank):
        try:
            capi.init(0)
            tr = RowPartitionedNGCFTrainer(ThreadComm(group, rank), U0, V0, W, adj, lr=0.002, reg=1e-3)
            lo, hi, pad = tr.rp.lo, tr.rp.hi, tr.rp.rows_pad
            losses = []
            for u, i, j, masks in steps:
                mine = [np.zeros((pad, ld), np.float32) for _ in masks]
                for m_blk, m in zip(mine, masks):
                    m_blk[:hi - lo] = m[lo:hi]
                tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B, masks=[DB.from_numpy(m) for m in mine])
                losses.append(tr.loss())
            inf = tr.inference_embeddings()
            capi.device_sync()
            result[rank] = (lo, hi, tr.block(tr.E[0]), tr.weights(), losses, inf)
        except Exception as e:      # noqa: BLE001
            errors.append(e); group.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    covered = 0
    for lo, hi, E_blk, Wr, losses, (Ur, Vr) in result:
        check_rel("logical ranks vs one GPU, losses", losses, losses_one, 1e-5)
        check("rel_err(E_blk, E_one[lo:hi])", rel_err(E_blk, E_one[lo:hi]), 1e-5)
        for k in range(2):
            for t in range(2):
                check("row-partitioned NGCF, logical ranks vs one GPU after two Adam steps: weights", rel_err(Wr[k][t], W1[k][t]), 2e-5, kind="partition")
        check("row-partitioned NGCF, logical ranks vs one GPU after two Adam steps: inference users", rel_err(Ur, Ui), 2e-5, kind="partition")
        check("row-partitioned NGCF, logical ranks vs one GPU after two Adam steps: inference items", rel_err(Vr, Vi), 2e-5, kind="partition")
        covered += hi - lo
    assert covered == N and not np.allclose(E_one[:nu], U0)


@pytest.mark.parametrize("world,layers", [(2, 2), (3, 1), (2, 3)])
@pytest.mark.parametrize("exchange", ["allgather", "referenced"])
def test_row_partitioned_simgcl_step_equals_the_single_gpu_step(world, layers, exchange, monkeypatch):
    """RowPartitionedSimGCLTrainer with G logical ranks against SimGCLTrainer on one GPU: same batch, same injected noise
    (each rank is handed its rows of it), same unique-row lists.  BPR and InfoNCE losses, the rank's rows of E after three
    steps and the clean encoder's embeddings agree to fp32 summation order."""
    monkeypatch.setenv("QREC_GRAPH_EXCHANGE", exchange)      # referenced: only the remote rows a rank's block refers to travel (round 3)
    from qrec_amd.graph import RowPartitionedSimGCLTrainer, SimGCLTrainer, joint_norm_adjacency, unique_first_appearance
    d = make_dataset("small")
    nu, ni, dim, B = d["n_users"], d["n_items"], 64, 512
    N = nu + ni
    adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
    rng = np.random.default_rng(30 + world)
    lim = np.sqrt(6.0 / (nu + dim))
    U0 = rng.uniform(-lim, lim, (nu, dim)).astype(np.float32); V0 = rng.uniform(-lim, lim, (ni, dim)).astype(np.float32)
    steps = []
    for _ in range(3):
        sel = rng.integers(0, d["train_u"].size, B)
        u = d["train_u"][sel].astype(np.int32); i = d["train_i"][sel].astype(np.int32); j = rng.integers(0, ni, B).astype(np.int32)
        steps.append((u, i, j, unique_first_appearance(u), (unique_first_appearance(i) + nu).astype(np.int32),
                      [rng.random((N, dim)).astype(np.float32) for _ in range(2 * layers)]))
    hp = dict(lr=0.001, reg=1e-4, cl_rate=0.5, eps=0.1, max_unique=B)
    one = SimGCLTrainer(U0, V0, adj, layers, **hp)
    losses_one = []
    for u, i, j, uu, vv, noises in steps:
        one.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B, DB.from_numpy(uu), uu.size, DB.from_numpy(vv), vv.size,
                             noises=[DB.from_numpy(x) for x in noises])
        losses_one.append(one.losses())
    E_one = np.concatenate(one.ego_embeddings())
    Um, Vm = one.main_embeddings()
    group = _Group(world)
    result, errors = [None] * world, []

    def run(rank):
        try:
            capi.init(0)
            tr = RowPartitionedSimGCLTrainer(ThreadComm(group, rank), U0, V0, adj, layers, **hp)
            lo, hi, pad = tr.rp.lo, tr.rp.hi, tr.rp.rows_pad
            losses = []
            for u, i, j, uu, vv, noises in steps:
                mine = []
                for x in noises:
                    blk = np.zeros((pad, dim), np.float32); blk[:hi - lo] = x[lo:hi]; mine.append(DB.from_numpy(blk))
                tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B, DB.from_numpy(uu), uu.size, DB.from_numpy(vv), vv.size,
                                    noises=mine)
                losses.append(tr.losses())
            emb = tr.main_embeddings()
            capi.device_sync()
            result[rank] = (lo, hi, tr.block(tr.E), losses, emb)
        except Exception as e:      # noqa: BLE001
            errors.append(e); group.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    covered = 0
    for lo, hi, E_blk, losses, (Ur, Vr) in result:
        check_rel("logical ranks vs one GPU, losses (b)", np.array(losses), np.array(losses_one), 1e-5)
        check("rel_err(E_blk, E_one[lo:hi])", rel_err(E_blk, E_one[lo:hi]), 1e-5)
        check("rel_err(Ur, Um)", rel_err(Ur, Um), 1e-5)
        check("rel_err(Vr, Vm)", rel_err(Vr, Vm), 1e-5)
        covered += hi - lo
    assert covered == N and not np.allclose(E_one[:nu], U0)


@pytest.mark.parametrize("pipeline,plan_ahead", [(False, ""), (True, ""), (False, "ahead"), (True, "ahead"), (False, "inside"), (True, "inside")])
@pytest.mark.parametrize("world,n_batches,dim,same_users", [(2, 3, 16, False), (3, 2, 64, False), (1, 2, 16, False), (2, 4, 64, True)])
def test_sharded_item_table_with_logical_ranks_equals_the_definition(world, n_batches, dim, same_users, pipeline, plan_ahead):
    """ShardedItemExchange + the real kernels, G logical ranks on one device.  The SGD kernel runs with ONE group, where
    it is the sequential recurrence, so the whole protocol is deterministic and comparable to the oracle.  ``same_users``:
    the weak-scaling layout -- every rank its own population with the SAME interaction structure, so all ranks ask the
    owners for the same positive rows in the same batches and every returned row arrives once per rank.
    ``pipeline`` (round 3): batch b + 1 is fetched on a second stream / communicator under batch b's SGD, after the owners applied
    batch b - 1 and before they apply batch b: the definition with the table as of two batches back inside an epoch.
    ``plan_ahead`` (round 3): the plan of epoch k + 1 (distinct rows per owner, rewritten ids, the id exchange) runs on a third
    stream / communicator while epoch k trains ("ahead"), or its device part in front of epoch k's last batch on the training
    stream, the row counts read back behind an event ("inside", what bench.py runs) -- into the other slot of the plan buffers:
    same epochs, planned earlier."""
    from qrec_amd.engine import padded_ld
    d, indptr, ind, P0, Q0 = _tiny_problem(dim)
    U, I = d["n_users"], d["n_items"]
    ld = padded_ld(dim, np.float32)
    lr, ru, ri = 0.05, 0.01, 0.02
    group, group_f, group_p = _Group(world), _Group(world), _Group(world)
    n_steps = 3 if plan_ahead else 2
    result, errors = [None] * world, []

    def pad(a):
        out = np.zeros((a.shape[0], ld), np.float32); out[:, :a.shape[1]] = a
        return out

    def run(rank):
        try:
            capi.init(0)
            comm = ThreadComm(group, rank)
            lo, hi = (0, U) if same_users else user_block(U, world, rank)
            d_P = DB.from_numpy(pad((P0[lo:hi] * (1 + 0.1 * rank * same_users)).astype(np.float32)))
            d_Q = DB.from_numpy(pad(qd.shard_item_rows(Q0, world, rank)))
            ex = qd.ShardedItemExchange(comm, I, ld, d_Q, pipeline=(ThreadComm(group_f, rank), capi.Stream()) if pipeline else None,
                                        plan_ahead=(ThreadComm(group_p, rank), capi.Stream()) if plan_ahead == "ahead" else None)
            d_loss = DB.zeros(1, np.float64)
            steps = []
            for step in range(n_steps):
                u, li, j = _rank_triplets(indptr, ind, lo, hi, I, 100 * step + rank, "user")
                steps.append((u.size, DB.from_numpy(u), DB.from_numpy(li), DB.from_numpy(j)))
            uploaded = capi.Event(); uploaded.record(None)
            for step, (n_t, d_u, d_i, d_j) in enumerate(steps):
                ex.plan_epoch(d_i, d_j, n_t, n_batches)           # plan_ahead, step >= 1: adopts the plan made under / inside the previous epoch
                nxt = None
                if plan_ahead == "inside" and step + 1 < n_steps:
                    nxt = dict(d_i=steps[step + 1][2], d_j=steps[step + 1][3], n=steps[step + 1][0], n_batches=n_batches, after=uploaded)
                ex.run_epoch(lambda t0, nb, cache, rows, ci, cj, st: capi.bpr_sgd_hogwild(
                    d_P, cache, dim, ld, d_u.ptr + 4 * t0, ci, cj, nb, 32, 1, lr, ru, ri, d_loss, capi.HW_ATOMIC, st,
                    p_rows=hi - lo, q_rows=rows), next_epoch=nxt)
                if plan_ahead == "ahead" and step + 1 < n_steps:
                    ex.plan_epoch_ahead(steps[step + 1][2], steps[step + 1][3], steps[step + 1][0], n_batches, uploaded)
                if plan_ahead and step + 1 < n_steps:
                    assert ex._ahead is not None and ex._ahead["slot"] != ex._slot and ex._ahead["finished"] == (plan_ahead == "ahead")
            capi.device_sync()
            result[rank] = (lo, hi, d_P.numpy()[:, :dim], d_Q.numpy()[:, :dim], float(d_loss.numpy()[0]), ex.bytes_moved)
        except Exception as e:      # noqa: BLE001 -- a failing rank must not leave the others at the barrier
            errors.append(e); group.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    # single-process statement
    Q = Q0.astype(np.float64); loss = 0.0
    Ps = [P0 * (1 + 0.1 * r * same_users) for r in range(world)]          # weak layout: each rank its own user rows
    for step in range(n_steps):
        work = []
        for r in range(world):
            lo, hi = (0, U) if same_users else user_block(U, world, r)
            u, li, j = _rank_triplets(indptr, ind, lo, hi, I, 100 * step + r, "user")
            work.append((r, lo, hi, u, li, j, -(-u.size // n_batches)))
        waiting = []
        for b in range(n_batches):
            deltas = np.zeros_like(Q)
            for r, lo, hi, u, li, j, per in work:
                t0, t1 = min(b * per, u.size), min((b + 1) * per, u.size)
                if t1 == t0:
                    continue
                items = np.unique(np.concatenate([li[t0:t1], j[t0:t1]]))
                slot = np.full(I, -1, np.int32); slot[items] = np.arange(items.size, dtype=np.int32)
                cache = Q[items].copy(); Pr = Ps[r][lo:hi].copy()
                loss += O.bpr_sgd(Pr, cache, u[t0:t1], slot[li[t0:t1]], slot[j[t0:t1]], lr, ru, ri)
                Ps[r][lo:hi] = Pr
                deltas[items] += cache - Q[items]
            if pipeline:        # batch b is applied only after batch b + 1 has been fetched
                waiting.append(deltas)
                if len(waiting) == 2:
                    Q += waiting.pop(0)
            else:
                Q += deltas
        for dlt in waiting:
            Q += dlt
    got_loss = 0.0
    for r in range(world):
        lo, hi, Pr, Qr, l, moved = result[r]
        check("rel_err(Pr, Ps[r][lo:hi])", rel_err(Pr, Ps[r][lo:hi]), 1e-5)
        check("rel_err(Qr, Q[r::world])", rel_err(Qr, Q[r::world]), 1e-5)
        assert (moved > 0) == (world > 1)
        got_loss += l
    check("abs(got_loss - loss) / loss", abs(got_loss - loss) / loss, 1e-5)
    assert rel_err(Q, Q0.astype(np.float64)) > 1e-3


# ---- bench.py's N > 1 paths on the one device ----------------------------------------------------------------------
def _bench(args, env_extra, nproc=None, port=29541, timeout=900):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    return json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])


def test_two_rank_bench_path_on_one_device(tmp_path):
    """replicated item table, two real processes (both on device 0, staged gloo transport): users are sharded, the
    item-table replicas must be IDENTICAL after every epoch's delta all-reduce, the loss terms ride in the same
    collective so both device-side drivers log the same losses and take the same learning-rate decisions, and the
    user tables differ (each rank trains its own users)."""
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs-per-step", "3", "--no-cpu-baseline", "--shape", "ml1m", "--scaling", "weak"],
                 {"QREC_DIST_TEST_ONE_DEVICE": "1", "QREC_DIST_TEST_DUMP": str(tmp_path)}, nproc=2)
    assert out["n_gpus"] == 2 and "INVALID_AS_BENCH" in out and out["value"] > 0 and out["scaling"] == "weak"
    assert "12080x3706" in out["config"]["workload"] and "weak scaling" in out["config"]["workload"]        # labelled with its aggregate shape
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["Q"], r1["Q"])                                   # replicas reconciled exactly
    assert not np.array_equal(r0["P"], r1["P"])                               # different user shards
    np.testing.assert_array_equal(r0["log"][:, :2], r1["log"][:, :2])         # same loss, same lr on both ranks
    assert r0["log"].shape[0] == 3 and float(r0["lr"]) == float(r1["lr"])        # the last step's epochs (every step restarts)
    assert r0["log"][-1, 0] < r0["log"][0, 0]                                 # the summed loss goes down (not every step: the bold driver may halve first)


def test_bench_typed_with_gpus_2_starts_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 2 ...` exactly as the driver types it for N = 1 -- no launcher, no WORLD_SIZE: bench.py re-executes
    itself under torch.distributed.run with two ranks and rank 0's ONE JSON line is what the caller reads on stdout.  (One-device
    test hook: both ranks on device 0 over the staged transport.)  The layout is steered by environment here, as a caller with a
    fixed command line would."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QREC_DIST_TEST_ONE_DEVICE="1", QREC_DIST_TEST_DUMP=str(tmp_path), QREC_DIST_MODE="replicated")
    env.pop("QREC_SCALING", None)            # the default answers BASELINE.json's question: the ONE problem over N GPUs
    run = subprocess.run(["python3", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs-per-step", "2", "--no-cpu-baseline",
                          "--shape", "ml1m", "--recall-dataset", "lastfm", "--recall-epochs", "10", "--config4-triplets", "150000"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                             # ONE line on stdout, whatever the launcher and RCCL print
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["dist_mode"] == "replicated"
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["rccl_ranks_agree"] and len(mg["kernel_ms_per_rank"]["all"]) == 2
    assert mg["kernel_ms_per_rank"]["min"] > 0 and mg["kernel_ms_per_rank"]["max"] >= mg["kernel_ms_per_rank"]["min"]
    assert sum(mg["triplets_per_epoch_per_rank"]) == out["config"]["triplets_per_epoch_per_gpu"] + mg["triplets_per_epoch_per_rank"][1]
    assert mg["collectives_per_epoch"]["all_reduce"] == 1 and mg["collectives_per_epoch"]["payload_bytes_per_rank"] > 3706 * 64 * 4    # two ranks: the epoch close's fused all-reduce alone (dist.reconciliations_per_epoch)
    assert (tmp_path / "rank0.npz").exists() and (tmp_path / "rank1.npz").exists()
    # round 4: strong scaling is the default and the workload string says what is being scaled; the weak-scaling figure stands next
    # to it under its aggregate shape; the collectives' bytes come with the link-time arithmetic; and the metric's second half --
    # Recall@20 against order-exact training of the whole problem, trained through this very communicator -- is on the line
    assert "strong scaling" in out["config"]["workload"] and "6040x3706" in out["config"]["workload"]
    assert out["config"]["triplets_per_epoch_job"] == sum(mg["triplets_per_epoch_per_rank"])
    ws = out["weak_scaling"]
    assert ws["value"] > 0 and "12080x3706" in ws["workload"]
    pl = mg["predicted_link_ms_per_epoch"]
    assert pl["one_ring_153GBps"] > pl["seven_rings_1071GBps"] > 0
    assert mg["collectives_per_epoch"]["ring_wire_bytes_per_rank"] == mg["collectives_per_epoch"]["payload_bytes_per_rank"]     # 2 (G - 1) / G x payload x 1 sync at G = 2
    # round 5: first contact with the communicator is a checked, timed-out preflight; and the line carries north_star's layout -- a (here
    # scaled-down) share of config #4 per rank, item table row-sharded, per-batch all-to-all -- with bytes moved, time per rank, and the
    # layout's own paired Recall@20
    assert mg["preflight"]["ok"] and mg["preflight"]["world"] == 2 and mg["preflight"]["alltoall_rows_sent"] == 3
    c4 = out["other_configs"]["config4_sharded"]
    assert "row-sharded x2" in c4["workload"] and c4["batches_per_epoch"] >= 1 and c4["bytes_leaving_all_ranks_per_epoch"] > 0 and c4["ms_per_epoch"] > 0
    assert len(c4["epoch_ms_per_rank_by_events"]["all"]) == 2 and c4["triplet_updates_per_s_job"] == pytest.approx(2 * 150000 / (c4["ms_per_epoch"] * 1e-3))
    rs = c4["recall_at_20_of_the_layout"]
    assert rs["layout"] == "sharded" and rs["ranks"] == 2 and rs["dataset"] == "lastfm" and 0.05 < rs["recall"] < 0.2
    rc = out["recall_at_20"]
    assert rc["dataset"] == "lastfm" and rc["ranks"] == 2 and rc["layout"] == "replicated" and rc["epochs"] == 10 and rc["bar"] == 0.002
    assert 0.05 < rc["recall_exact_order"] < 0.2 and 0.05 < rc["recall"] < 0.2 and rc["abs_diff"] == pytest.approx(abs(rc["recall"] - rc["recall_exact_order"]))
    assert rc["rel_diff"] == pytest.approx(rc["abs_diff"] / rc["recall_exact_order"]) and rc["final"]["epoch"] == 10
    check("bench.py --gpus 2 (one device, staged transport): |Recall@20 - exact-order| at the peak epoch, lastfm, lr0 0.01", rc["abs_diff"], 0.002, inclusive=True, kind="statistical")


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_bench_path_with_row_sharded_item_table(tmp_path, scaling):
    """--dist-mode sharded: each rank holds half of the item rows; the epoch's batches fetch and return rows through
    the all-to-all exchange.  Both drivers see the same all-reduced loss terms; the run trains (loss goes down) and
    the two shards together are a table that moved on every row the epoch touched."""
    # weak layout: both ranks' populations hit the SAME hot items in the same batch; with the item-major schedule every
    # chunk of a hot item reads the batch-start row, so this case runs user-major; strong runs item-major
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--epochs-per-step", "3", "--no-cpu-baseline", "--shape", "ml1m",
                  "--dist-mode", "sharded", "--shard-batch", "250000", "--scaling", scaling,
                  "--schedule", "user" if scaling == "weak" else "item"] + (["--shard-pipeline"] if scaling == "strong" else []),
                 {"QREC_DIST_TEST_ONE_DEVICE": "1", "QREC_DIST_TEST_DUMP": str(tmp_path)}, nproc=2, port=29543)
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] > 0
    assert out["config"]["batches_per_epoch"] >= (2 if scaling == "strong" else 4) and out["config"]["xgmi_bytes_per_epoch_all_ranks"] > 0
    assert out["config"]["fetch_pipelined"] == (scaling == "strong") and out["config"]["plan"].startswith("inside")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    I = 3706
    assert r0["Q"].shape[0] == (I + 1) // 2 and r1["Q"].shape[0] == I // 2    # items 0,2,4.. / 1,3,5..
    np.testing.assert_array_equal(r0["log"][:, :2], r1["log"][:, :2])
    assert np.isfinite(r0["Q"]).all() and np.isfinite(r1["Q"]).all()
    assert r0["log"][-1, 0] < r0["log"][0, 0]
    Q0 = (np.random.default_rng(999).random((I, 64)) / 3).astype(np.float32)
    assert (np.abs(r0["Q"][:, :64] - Q0[0::2]).max(1) > 0).mean() > 0.99 and (np.abs(r1["Q"][:, :64] - Q0[1::2]).max(1) > 0).mean() > 0.99


@pytest.mark.parametrize("mode", ["replicated", "sharded"])
def test_bench_multi_gpu_path_on_real_rccl_world_one(mode):
    """QREC_FORCE_DIST=1: the N > 1 code path on the real RCCL communicator (world 1): delta / apply kernels or the
    all-to-all exchange with itself.  Must train like the plain single-GPU path does (same seeds, same kernels): the
    loss after the same number of epochs agrees closely (Hogwild timing, and for the sharded layout the batch-start
    snapshots of the item rows, aside)."""
    a = _bench(["--steps", "2", "--warmup", "1", "--epochs-per-step", "3", "--no-cpu-baseline", "--no-extras", "--shape", "ml1m",
                "--dist-mode", mode], {"QREC_FORCE_DIST": "1", "MASTER_PORT": "29547"})
    b = _bench(["--steps", "2", "--warmup", "1", "--epochs-per-step", "3", "--no-cpu-baseline", "--no-extras", "--shape", "ml1m"], {})
    assert a["n_gpus"] == 1 and "FORCE_DIST" in a["config"]["parallelism"]
    assert a["config"]["final_loss"] == pytest.approx(b["config"]["final_loss"], rel=0.03)
    # (the bold driver's decisions are usually the same ones; on this structureless shape the first epochs' losses differ by 1e-3, so a
    # Hogwild-timing-sized difference may flip ONE x1.05 / x0.5 decision of the three -- seen once in round 4 -- without moving the loss level)
    ratio = a["config"]["final_lr"] / b["config"]["final_lr"]
    assert ratio == pytest.approx(1.0) or ratio == pytest.approx(0.5 / 1.05) or ratio == pytest.approx(1.05 / 0.5)


# ---- graph models: 1-D row partition of the propagation --------------------------------------------
@pytest.mark.parametrize("world,layers,dim", [(3, 2, 16), (2, 3, 64), (8, 2, 50)])
@pytest.mark.parametrize("exchange", ["allgather", "referenced"])
def test_row_partitioned_lightgcn_step_equals_the_single_gpu_step(world, layers, dim, exchange, monkeypatch):
    """RowPartitionedLightGCNTrainer with G logical ranks (threads + in-process collective): every rank holds its rows of
    the adjacency, of E and of the Adam slots; layers are all-gathered.  Same batch, same step as LightGCNTrainer on
    one GPU: the propagated layer sum is bit-identical row for row (same SpMM kernel, same segment order), the tables
    after two steps agree to the summation order of the batch gradient's float atomics."""
    monkeypatch.setenv("QREC_GRAPH_EXCHANGE", exchange)      # referenced: only the remote rows a rank's block refers to travel (round 3)
    from qrec_amd.graph import LightGCNTrainer, RowPartitionedLightGCNTrainer, joint_norm_adjacency
    d = make_dataset("small")
    nu, ni = d["n_users"], d["n_items"]
    adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
    rng = np.random.default_rng(world)
    U0 = (rng.standard_normal((nu, dim)) * 0.1).astype(np.float32); V0 = (rng.standard_normal((ni, dim)) * 0.1).astype(np.float32)
    B = 512
    batches = [(rng.integers(0, nu, B).astype(np.int32), rng.integers(0, ni, B).astype(np.int32), rng.integers(0, ni, B).astype(np.int32))
               for _ in range(2)]
    one = LightGCNTrainer(U0, V0, adj, layers, lr=0.01, reg=1e-3)
    one.forward_sum()
    S_one = one.S.numpy()[:, :dim].copy()
    losses_one = []
    for u, i, j in batches:
        one.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B); losses_one.append(one.loss())
    E_one = one.E.numpy()[:, :dim]
    group = _Group(world)
    result, errors = [None] * world, []

    def run(rank):
        try:
            capi.init(0)
            tr = RowPartitionedLightGCNTrainer(ThreadComm(group, rank), U0, V0, adj, layers, lr=0.01, reg=1e-3)
            tr.forward_sum(); S_blk = tr.block(tr.S)
            losses = []
            for u, i, j in batches:
                tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B); losses.append(tr.loss())
            capi.device_sync()
            result[rank] = (tr.rp.lo, tr.rp.hi, S_blk, tr.block(tr.E), losses)
        except Exception as e:      # noqa: BLE001
            errors.append(e); group.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    covered = 0
    for lo, hi, S_blk, E_blk, losses in result:
        assert np.array_equal(S_blk, S_one[lo:hi])                         # forward propagation: the same bits
        np.testing.assert_allclose(E_blk, E_one[lo:hi], rtol=0, atol=3e-6)
        check_rel("row-partitioned LightGCN vs one GPU, losses", losses, losses_one, 2e-6)
        covered += hi - lo
    assert covered == nu + ni and not np.allclose(E_one[:nu], U0)
