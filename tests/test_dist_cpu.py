"""world_size-2 tests of the multi-GPU schemes on CPU (gloo).  The orchestration under test is the product's
(qrec_amd/dist.py: ReplicatedTableSync, ShardedItemExchange, BatchParallel, RowPartition); the kernels it enqueues are
replaced by tests/hostkern.py (numpy, same entry points) and the SGD by the oracle, so that what is checked here is the
protocol: user sharding is a partition, and a sharded step equals the single-process definition
    replicated:  Q_start + sum_r (Q_r - Q_start)                    per epoch
    sharded:     Q + sum_r (cache_r_after - cache_r_before)         per batch, rows fetched from / returned to their owners."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c as O
from qrec_amd import dist as qd
from qrec_amd.dist import ReplicatedTableSync, shard_positive_csr, user_block
from qrec_amd.synth import make_dataset, to_csr
from tests import hostkern as HK

D, LD = 16, 32


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _problem():
    d = make_dataset("tiny")
    indptr, ind = to_csr(d["n_users"], d["train_u"], d["train_i"])
    rng = np.random.default_rng(0)
    P0 = rng.random((d["n_users"], D)) / 3; Q0 = (rng.random((d["n_items"], D)) / 3).astype(np.float32)
    return d, indptr, ind, P0, Q0


def _pad(Q):
    out = np.zeros((Q.shape[0], LD), np.float32); out[:, :Q.shape[1]] = Q
    return out


def _shard(indptr, ind, lo, hi, n_items, seed):
    lp, li = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64), np.ascontiguousarray(ind[indptr[lo]:indptr[hi]])
    u = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(lp)).astype(np.int32)
    j = O.bpr_sample_epoch(O.MT.cpython_seed(seed), lp, li, n_items)
    return u, li, j


def _sgd_on(P, Qpad, u, i, j):
    """order-exact oracle SGD on (P fp64, a [rows][LD] fp32 table or row cache): the rank's 'kernel'"""
    Q = Qpad[:, :D].astype(np.float64)
    O.bpr_sgd(P, Q, np.ascontiguousarray(u, np.int32), np.ascontiguousarray(i, np.int32), np.ascontiguousarray(j, np.int32), 0.05, 0.01, 0.01)
    Qpad[:, :D] = Q.astype(np.float32)


def _local_epoch(indptr, ind, lo, hi, P0, Qpad, n_items, seed):
    """one rank's work: its users' triplets, own sampler stream, order-exact on its replica"""
    u, li, j = _shard(indptr, ind, lo, hi, n_items, seed)
    P = P0[lo:hi].copy()
    _sgd_on(P, Qpad, u, li, j)
    return P


def _join(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    control = qd.ControlPlane.from_env()
    return control, qd.GlooStagedComm(control, kern=HK)


def _worker(rank, world, port, out):
    control, comm = _join(rank, world, port)
    d, indptr, ind, P0, Q0 = _problem()
    lo, hi, lp, li = shard_positive_csr(indptr, ind, world, rank)
    q = HK.DeviceBuffer.from_numpy(_pad(Q0))
    stats = HK.DeviceBuffer.from_numpy(np.array([1.0 + rank, 10.0 * (rank + 1)]))
    sync = ReplicatedTableSync(comm, q, kern=HK)
    P = None
    for step in range(2):
        P0_step = P0 if P is None else np.concatenate([P0[:lo], P, P0[hi:]])
        P = _local_epoch(indptr, ind, lo, hi, P0_step, q.a, d["n_items"], 100 * step + rank)
        sync.sync(extra=(stats, 2, HK.F64) if step == 1 else None)
    out[rank] = (lo, hi, P, q.a.copy(), stats.a.copy())
    control.shutdown()


def _hot_worker(rank, world, port, out):
    """K = 3 batches per epoch: two INNER reconciliations, then the full one (what engine.epoch_device_async enqueues for the replicated
    layout); the 'SGD' of a batch is a rank- and batch-specific additive change"""
    control, comm = _join(rank, world, port)
    rng = np.random.default_rng(0)
    n_rows, ld = 300, 8
    Q0 = rng.standard_normal((n_rows, ld)).astype(np.float32)
    q = HK.DeviceBuffer.from_numpy(Q0)
    step = qd.ReplicatedStep(comm, ReplicatedTableSync(comm, q, kern=HK))
    mid = []
    for b in range(3):
        q.a += _batch_change(rank, b, n_rows, ld)
        if b < 2:
            step.sync_tables()
            mid.append(q.a.copy())
        else:
            step.sync_q.sync()
    out[rank] = (q.a.copy(), mid)
    control.shutdown()


def _preflight_worker(rank, world, port, out):
    control, comm = _join(rank, world, port)
    out[rank] = qd.preflight(comm, kern=HK, timeout_s=60)
    control.shutdown()


def _diverging_worker(rank, world, port, out):
    control, _ = _join(rank, world, port)
    got = {}
    assert control.allreduce_host(np.array([1.0 + rank]))[0] == 3.0                  # the same collective from the same site: nothing said
    try:                                                                            # rank 0 gathers, rank 1 is already at a barrier
        if rank == 0:
            control.allgather_host(np.zeros(3))
        else:
            control.barrier()
        got["diverged"] = None
    except RuntimeError as e:
        got["diverged"] = str(e)
    # the same collective from two different call sites is legal (it completes): said once on stderr, never raised
    if rank == 0:
        a = control.allreduce_host(np.array([2.0]))
    else:
        a = control.allreduce_host(np.array([5.0]))
    got["two_sites"] = float(a[0]); got["warned"] = control._warned_sites
    out[rank] = got
    control.shutdown()


def test_ranks_in_different_collectives_raise_instead_of_hanging():
    """round 5: `bench.py --gpus 2` hung for its whole 900 s limit because a collective leg sat behind a condition only rank 0 satisfied --
    rank 0 waited in an all-gather, rank 1 at the final barrier.  The control plane now exchanges (operation, call site) tags in front of every
    collective: ranks in DIFFERENT collectives raise on every rank at once, each rank's operation and site in the message."""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_diverging_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        msg = out[r]["diverged"]
        assert msg is not None and "not in the same collective" in msg
        assert "rank 0: allgather_host at test_dist_cpu.py:_diverging_worker" in msg and "rank 1: barrier at test_dist_cpu.py:_diverging_worker" in msg
        assert out[r]["two_sites"] == 7.0 and out[r]["warned"] is True


def test_preflight_round_trip_over_two_processes():
    """dist.preflight (round 5: what `bench.py --gpus N` and the drop-in classes run before anything is timed) over two gloo processes: the
    tiny all-reduce and the ragged all-to-all round trip return what they must; both ranks report the same all-reduce share of the checksum"""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_preflight_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0]["ok"] and out[1]["ok"] and out[0]["world"] == 2 and out[0]["allreduce_floats"] == 256
    assert out[0]["alltoall_rows_sent"] == (1 + 0) + (1 + 1) and out[1]["alltoall_rows_sent"] == (1 + 1) + (1 + 2)


def test_preflight_names_a_wrong_collective_and_a_hung_one():
    """... a transport that loses the all-to-all's payload is named (collective, rank, row); one that never returns trips the watchdog
    with a one-line reason instead of hanging the job"""
    import time

    class Alone:
        world, rank = 1, 0
        def allreduce(self, *a, **k): pass                                     # world 1: the identity IS right
        def alltoall_rows(self, send, s_rows, recv, r_rows, row_bytes, stream=None):
            HK.memcpy_h2d(recv, send.a, send.nbytes)

    assert qd.preflight(Alone(), kern=HK, timeout_s=30)["ok"]

    class Lossy(Alone):
        def alltoall_rows(self, *a, **k): pass
    with pytest.raises(RuntimeError, match="preflight all-to-all: rank 0 of 1 received"):
        qd.preflight(Lossy(), kern=HK, timeout_s=30)

    class Wrong(Alone):
        world = 2                                                              # claims two ranks, sums one
    with pytest.raises(RuntimeError, match="preflight all-reduce: rank 0 of 2 got"):
        qd.preflight(Wrong(), kern=HK, timeout_s=30)

    class Hung(Alone):
        def allreduce(self, *a, **k): time.sleep(5)
    said = []
    r = qd.preflight(Hung(), kern=HK, timeout_s=0.3, on_hang=said.append)
    assert r["ok"] is False and len(said) == 1 and "did not complete within" in said[0] and "\n" not in said[0]
    # the same guard around a blocking start-up call (bench.py wraps the communicator's creation in it)
    said = []
    with qd.watchdog("creating the communicator", 0.2, on_hang=said.append):
        time.sleep(0.6)
    with qd.watchdog("something quick", 5.0, on_hang=said.append):
        pass
    time.sleep(0.1)
    assert len(said) == 1 and "creating the communicator did not complete within" in said[0]


def _batch_change(rank, b, n_rows, ld):
    return np.random.default_rng(1000 * rank + b).standard_normal((n_rows, ld)).astype(np.float32) * 0.1


def test_inner_reconciliations_count_nothing_twice():
    """engine.epoch_device_async, replicated layout with K = 3 batches: two inner reconciliations of the whole table and the epoch's full
    one.  Afterwards every replica equals start + the sum of EVERY rank's change of EVERY batch (no inner delta is added again by the
    full sync); in between the replicas agree."""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_hot_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    n_rows, ld = 300, 8
    want = rng.standard_normal((n_rows, ld)).astype(np.float32).astype(np.float64)
    for r in range(world):
        for b in range(3):
            want += _batch_change(r, b, n_rows, ld)
    (q0, mid0), (q1, mid1) = out[0], out[1]
    assert np.array_equal(q0, q1)
    np.testing.assert_allclose(q0, want, rtol=0, atol=2e-6)
    for a, b in zip(mid0, mid1):
        assert np.array_equal(a, b)


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


def test_item_rows_are_partitioned_interleaved():
    Q = np.arange(23 * 2, dtype=np.float32).reshape(23, 2)
    for w in (1, 2, 3, 8):
        parts = [qd.shard_item_rows(Q, w, r) for r in range(w)]
        assert sum(p.shape[0] for p in parts) == 23
        assert [p.shape[0] for p in parts] == [HK.shard_rows(23, w, r) for r in range(w)]
        owner, row = qd.item_owner(np.arange(23), w)
        for it in range(23):
            assert np.array_equal(parts[owner[it]][row[it]], Q[it])


def test_two_rank_step_equals_definition():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d, indptr, ind, P0, Q0 = _problem()
    # single-process statement of the same semantics
    Q = _pad(Q0); P = P0.copy()
    for step in range(2):
        deltas = []
        newP = P.copy()
        for r in range(world):
            lo, hi = user_block(d["n_users"], world, r)
            Qr = Q.copy()
            newP[lo:hi] = _local_epoch(indptr, ind, lo, hi, P, Qr, d["n_items"], 100 * step + r)
            deltas.append(Qr - Q)
        Q = Q + sum(deltas); P = newP
    assert np.array_equal(out[0][3], out[1][3])                          # replicas bit-identical
    for r in range(world):
        lo, hi, Pr, Qr, stats = out[r]
        np.testing.assert_allclose(Qr, Q, rtol=0, atol=1e-7)             # ... and equal to the definition
        np.testing.assert_allclose(Pr, P[lo:hi], rtol=1e-6, atol=1e-8)
        assert np.array_equal(stats, [3.0, 30.0])                        # the loss terms rode in the same collective
    assert not np.allclose(Q, _pad(Q0))


# ---- row-sharded item table: all-to-all row lookups + return of the updates ------------------------------------------
N_BATCHES = 3


def _rank_users(n_users, world, rank, same_users):
    """strong-scaling split (disjoint user blocks) or the weak-scaling layout (every rank its own population with the
    SAME interaction structure: all ranks ask the owners for the same positive rows in the same batches)"""
    return (0, n_users) if same_users else user_block(n_users, world, rank)


def _sharded_worker(rank, world, port, out, same_users=False, pipeline=False, plan_ahead=False):
    control, comm = _join(rank, world, port)
    d, indptr, ind, P0, Q0 = _problem()
    I = d["n_items"]
    lo, hi = _rank_users(d["n_users"], world, rank, same_users)
    q_local = HK.DeviceBuffer.from_numpy(_pad(qd.shard_item_rows(Q0, world, rank)))
    ex = qd.ShardedItemExchange(comm, I, LD, q_local, kern=HK, pipeline=(comm, None) if pipeline else None,
                                plan_ahead=(comm, None) if plan_ahead == "ahead" else None)
    assert ex.rows_local == q_local.a.shape[0]
    P = P0[lo:hi].copy()
    steps = []
    for step in range(3 if plan_ahead else 2):
        u, li, j = _shard(indptr, ind, lo, hi, I, 100 * step + rank)
        if rank == 1 and step == 1:            # ragged: this rank runs out of triplets before the others do
            u, li, j = u[:5], li[:5], j[:5]
        if rank == 1 and step == 2:            # ... and has none at all in the third epoch: it still plans, answers and applies
            u, li, j = u[:0], li[:0], j[:0]
        steps.append((u, HK.DeviceBuffer.from_numpy(li), HK.DeviceBuffer.from_numpy(j)))
    for step, (u, d_i, d_j) in enumerate(steps):
        ex.plan_epoch(d_i, d_j, u.size, N_BATCHES)          # plan_ahead: adopts the plan made under / inside the previous epoch
        assert ex._slot == (step + 1) % 2                   # the plans alternate between the two slots
        nxt = None
        if plan_ahead and step + 1 < len(steps):
            un, d_in, d_jn = steps[step + 1]
            if plan_ahead == "ahead":                       # (on the device this is enqueued behind run_epoch; the host emulation has no "behind")
                ex.plan_epoch_ahead(d_in, d_jn, un.size, N_BATCHES, HK.Event())
            else:                                           # "inside": begun by run_epoch in front of its last batch, finished by the next plan_epoch
                nxt = dict(d_i=d_in, d_j=d_jn, n=un.size, n_batches=N_BATCHES, after=HK.Event())

        def sgd_batch(t0, nb, cache, rows, ci, cj, stream):
            c = HK._view(cache, rows * LD, np.float32).reshape(rows, LD)
            _sgd_on(P, c, u[t0:t0 + nb], HK._view(ci, nb, np.int32), HK._view(cj, nb, np.int32))
        ex.run_epoch(sgd_batch, next_epoch=nxt)
        if nxt is not None:
            assert ex._ahead is not None and not ex._ahead["finished"] and ex._ahead["slot"] != ex._slot
    out[rank] = (lo, hi, P, q_local.a.copy(), ex.bytes_moved)
    control.shutdown()


@pytest.mark.parametrize("same_users,pipeline,plan_ahead", [(False, False, ""), (True, False, ""), (False, True, ""), (True, True, ""),
                                                             (False, False, "ahead"), (True, True, "ahead"), (False, False, "inside"), (True, True, "inside")])
def test_two_rank_sharded_item_table_equals_definition(same_users, pipeline, plan_ahead):
    """``pipeline``: the fetch of batch b + 1 is issued under batch b's SGD, ordered after the owners applied batch b - 1 and before
    they apply batch b -- a batch sees the table as of TWO batches back inside an epoch (everything at an epoch's start).
    ``plan_ahead``: every epoch's plan (distinct rows, id exchange) is made one epoch early, into the other slot of the plan
    buffers -- "ahead": all of it, on a plan stream / communicator; "inside": its device part in front of the previous epoch's
    last batch, its host part by the epoch's own ``plan_epoch`` -- which changes when the plan is computed, not what the epoch does."""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, _free_port(), out, same_users, pipeline, plan_ahead), nprocs=world, join=True)
    d, indptr, ind, P0, Q0 = _problem()
    I = d["n_items"]
    Q = _pad(Q0); Pr_all = [P0.copy() for _ in range(world)]       # weak layout: every rank its own copy of the user rows
    for step in range(3 if plan_ahead else 2):
        work = []
        for r in range(world):
            lo, hi = _rank_users(d["n_users"], world, r, same_users)
            u, li, j = _shard(indptr, ind, lo, hi, I, 100 * step + r)
            if r == 1 and step == 1:
                u, li, j = u[:5], li[:5], j[:5]
            if r == 1 and step == 2:
                u, li, j = u[:0], li[:0], j[:0]
            per = -(-u.size // N_BATCHES)
            work.append((r, lo, hi, u, li, j, per))
        waiting = []            # pipelined: batches fetched but not yet applied when the next fetch goes out (at most one)
        for b in range(N_BATCHES):
            deltas = np.zeros_like(Q)
            for r, lo, hi, u, li, j, per in work:
                t0, t1 = min(b * per, u.size), min((b + 1) * per, u.size)
                if t1 == t0:
                    continue
                items = np.unique(np.concatenate([li[t0:t1], j[t0:t1]]))
                slot = np.full(I, -1); slot[items] = np.arange(items.size)
                cache = Q[items].copy()                                   # the rows as the owners hold them when they answer
                Pr = Pr_all[r][lo:hi]
                _sgd_on(Pr, cache, u[t0:t1], slot[li[t0:t1]], slot[j[t0:t1]])
                deltas[items] += cache - Q[items]
            if pipeline:        # batch b is applied only after batch b + 1 has been fetched
                waiting.append(deltas)
                if len(waiting) == 2:
                    Q = Q + waiting.pop(0)
            else:
                Q = Q + deltas
        for dlt in waiting:
            Q = Q + dlt
    for r in range(world):
        lo, hi, Pr, Qr, moved = out[r]
        np.testing.assert_allclose(Qr, Q[r::world], rtol=0, atol=3e-7)    # every owner holds the definition's rows
        np.testing.assert_allclose(Pr, Pr_all[r][lo:hi], rtol=1e-6, atol=1e-8)
        assert moved > 0
    assert not np.allclose(Q, _pad(Q0))


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
    return T.LightGCN(U0, V0, A, 2, lr=0.01, reg=1e-3), u, i, j, A


def _graph_worker(rank, world, port, out):
    control, comm = _join(rank, world, port)
    qd.attach(control, comm)
    dp = qd.BatchParallel.from_env()
    model, u, i, j, _ = _graph_problem()
    steps = []
    for step in range(2):
        lo, cnt = dp.share(u.size)
        loss, g = model.loss_and_grad(u[lo:lo + cnt], i[lo:lo + cnt], j[lo:lo + cnt])     # this rank's share of the step
        g = dp.all_reduce_host(g); loss = float(dp.all_reduce_host(np.array([loss], np.float64))[0])
        model.opt.step(model.E, g)                                                          # same update on every replica
        steps.append(loss)
    out[rank] = (dp.share(u.size), steps, model.E.copy(), qd.is_output_rank())
    control.shutdown()


def test_graph_batch_parallel_equals_the_whole_step():
    """two ranks, each with a share of the step's rows + gradient all-reduce == one process on the whole step"""
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_graph_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    model, u, i, j, _ = _graph_problem()
    want = [model.train_step(u, i, j) for _ in range(2)]
    shares = [out[r][0] for r in range(world)]
    assert shares[0][0] == 0 and shares[0][1] + shares[1][1] == u.size and shares[1][0] == shares[0][1]
    assert np.array_equal(out[0][2], out[1][2])                         # replicas identical
    np.testing.assert_allclose(out[0][1], want, rtol=1e-5)
    np.testing.assert_allclose(out[0][2], model.E, rtol=0, atol=2e-5)
    assert out[0][3] is True and out[1][3] is False                     # rank 0 alone writes result files


# ---- graph models: 1-D row partition of the propagation (qrec_amd.dist.RowPartition) ------------------------------
def _rowpart_worker(rank, world, port, out, n_sub=None):
    control, comm = _join(rank, world, port)
    _, _, _, _, A = _graph_problem()
    if n_sub:
        A = A[:n_sub, :n_sub]
    n, ld = A.shape[0], 8
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, ld)).astype(np.float32); dY = rng.standard_normal((n, ld)).astype(np.float32)
    rp = qd.RowPartition(comm, n, ld, kern=HK)
    pad = rp.rows_pad
    blk = np.zeros((pad, ld), np.float32); blk[:rp.hi - rp.lo] = X[rp.lo:rp.hi]
    d_blk, d_full = HK.DeviceBuffer.from_numpy(blk), HK.DeviceBuffer((world * pad, ld), np.float32)
    rp.gather_operand(d_blk, d_full)                                     # forward: all-gather, then the rank's rows of A
    Y_rows = A[rp.lo:rp.hi] @ d_full.a[:n]
    part = np.zeros((world * pad, ld), np.float32)                       # backward: A[rows]^T dY[rows], reduce-scatter
    part[:n] = A[rp.lo:rp.hi].T @ dY[rp.lo:rp.hi]
    d_part, d_dx = HK.DeviceBuffer.from_numpy(part), HK.DeviceBuffer((pad, ld), np.float32)
    rp.scatter_sum(d_part, d_dx)
    # referenced rows only (round 3): the block's columns renumbered into [own rows | remote rows it refers to], the operand assembled
    # by one row gather + one row all-to-all -- the same product from fewer bytes
    A_csr = A.tocsr(); A_csr.sort_indices()
    cols = rp.reference(A_csr.indptr.astype(np.int64), A_csr.indices.astype(np.int32))
    d_cmp = HK.DeviceBuffer((rp.ref_rows, ld), np.float32)
    rp.gather_referenced(d_blk, d_cmp)
    ptr = A_csr.indptr[rp.lo:rp.hi + 1] - A_csr.indptr[rp.lo]
    vals = A_csr.data[A_csr.indptr[rp.lo]:A_csr.indptr[rp.hi]]
    import scipy.sparse as sp
    Y_ref = sp.csr_matrix((vals, cols, ptr), shape=(rp.hi - rp.lo, rp.ref_rows)) @ d_cmp.a
    out[rank] = (rp.lo, rp.hi, Y_rows, d_dx.a[:rp.hi - rp.lo].copy(), Y_ref, rp.ref_rows, rp.ref_bytes_per_product)
    control.shutdown()


def test_row_partitioned_propagation_equals_the_whole_product():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_rowpart_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    _, _, _, _, A = _graph_problem()
    n, ld = A.shape[0], 8
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, ld)).astype(np.float32); dY = rng.standard_normal((n, ld)).astype(np.float32)
    Y, dX = A @ X, A.T @ dY
    assert out[0][0] == 0 and out[0][1] == out[1][0] and out[1][1] == n
    for r in range(world):
        lo, hi, Yr, dXr, Yref, ref_rows, ref_bytes = out[r]
        np.testing.assert_allclose(Yr, Y[lo:hi], rtol=0, atol=1e-6)
        np.testing.assert_allclose(dXr, dX[lo:hi], rtol=0, atol=2e-6)
        np.testing.assert_allclose(Yref, Y[lo:hi], rtol=0, atol=1e-6)        # referenced-rows exchange: the same rows of the product
        # the compact operand never holds more than the whole table beside the rank's own (padded) block; the bytes a rank SENDS per
        # product are rows of its own block, at most once to every other rank
        assert ref_rows <= n + (hi - lo) + 32 and 0 < ref_bytes <= (hi - lo) * (world - 1) * ld * 4


def test_row_partition_with_empty_trailing_ranks():
    """rows per rank are rounded up to a multiple of 32 (bitmap word boundary), so on a table of fewer than 32 x world rows the
    trailing ranks own NOTHING (lo == hi == n_rows).  70 rows over 4 ranks = blocks of 32 / 32 / 6 / 0: every collective keeps its uniform
    shape, the empty rank contributes zeros, refers to nothing and is sent nothing, and the others' products are the whole product's rows."""
    world, n = 4, 70
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_rowpart_worker, args=(world, _free_port(), out, n), nprocs=world, join=True)
    _, _, _, _, A = _graph_problem()
    A = A[:n, :n]
    ld = 8
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, ld)).astype(np.float32); dY = rng.standard_normal((n, ld)).astype(np.float32)
    Y, dX = A @ X, A.T @ dY
    assert [(out[r][0], out[r][1]) for r in range(world)] == [(0, 32), (32, 64), (64, 70), (70, 70)]
    for r in range(world):
        lo, hi, Yr, dXr, Yref, ref_rows, ref_bytes = out[r]
        assert Yr.shape == (hi - lo, ld) and dXr.shape == (hi - lo, ld) and Yref.shape == (hi - lo, ld)
        np.testing.assert_allclose(Yr, Y[lo:hi], rtol=0, atol=1e-6)
        np.testing.assert_allclose(dXr, dX[lo:hi], rtol=0, atol=2e-6)
        np.testing.assert_allclose(Yref, Y[lo:hi], rtol=0, atol=1e-6)
    assert out[3][5] == 32 and out[3][6] == 0            # the empty rank: a compact operand of its own (all-pad) block, nothing sent
