"""Multi-GPU execution of the embedding-training hot path -- one process per GPU.

Data plane: RCCL over xGMI, bound directly by libqrec_hip.so (``capi.Comm``: qrec_comm_init / qrec_allreduce /
qrec_alltoall_rows ...) and enqueued on the same HIP stream as the kernels; no torch tensor or torch kernel is on it.
Control plane: ``torch.distributed`` with the gloo backend (TCP between the ranks' hosts) -- it hands out the RCCL id
and the seeds, and carries barriers and a few host integers; never embedding data.

BPR (config #4).  Users are split into contiguous id blocks, one per rank: a rank owns its users' rows of ``P``, their
triplets (the PositiveSet CSR is user-major, so this is a row split) and their negative sampling -- no exchange on that
side.  The item table ``Q`` has two layouts:

``replicated``  every rank holds all of ``Q`` and trains its triplets on its copy; after the SGD kernel the copies are
    reconciled by summing the ranks' deltas,  Q <- Q_start + sum_r (Q_r - Q_start)  (ReplicatedTableSync: delta kernel,
    ONE fused all-reduce that also carries the epoch's loss terms, apply kernel).  Every rank's updates are kept, a row
    read during a step lags the other ranks' updates by at most one step.  Traffic: |Q| floats per step and rank.
``sharded``     item ``r*G + o`` is local row ``r`` of rank ``o`` (BASELINE.json's north star: row-sharded tables, RCCL
    all-to-all for cross-shard lookups).  An epoch is cut into batches; per batch a rank asks the owners for the distinct
    item rows its triplets touch (all-to-all of row ids), receives them into a row cache (all-to-all of rows), runs the
    unchanged SGD kernel on (P, cache), and returns the cache; owners add ``returned - sent`` into their rows
    (ShardedItemExchange).  Same contract at batch granularity:  Q <- Q + sum_r (cache_r_after - cache_r_before).
    Memory per rank: |Q|/G + the cache of one batch.  Traffic: 2 rows per DISTINCT item a batch touches.
    Round 3: all batches of an epoch are planned in one set of launches with ONE id exchange; the next epoch's plan is begun
    in front of the current epoch's last batch and its row counts reach the host behind an event, so no epoch waits for a
    drained stream; optional: the fetch of batch k + 1 under batch k's SGD kernel (second stream + communicator), the whole
    plan on a third stream + communicator (ShardedItemExchange's ``pipeline`` / ``plan_ahead``).

Graph models (config #5): ``BatchParallel`` (batch-sharded steps, one gradient all-reduce) and ``RowPartition``
(1-D row partition of the propagation: all-gather of the operand per layer -- or, ``RowPartition.reference`` /
``gather_referenced``, a grouped send/recv of only the operand rows a rank's block of the adjacency refers to --,
reduce-scatter in the backward pass).
"""
from __future__ import annotations

import os

import numpy as np

from . import capi as _capi


def user_block(n_users: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of the contiguous user-id block owned by ``rank`` (sizes differ by <= 1)."""
    base, extra = divmod(n_users, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_positive_csr(indptr: np.ndarray, indices: np.ndarray, world: int, rank: int):
    """Row split of the user-major PositiveSet CSR.  Returns (lo, hi, local_indptr,
    local_indices): the rank's users keep their global item ids; user ids become local
    (u - lo) because the rank only stores its own rows of P."""
    lo, hi = user_block(indptr.size - 1, world, rank)
    b, e = int(indptr[lo]), int(indptr[hi])
    return lo, hi, (indptr[lo:hi + 1] - indptr[lo]).astype(np.int64), np.ascontiguousarray(indices[b:e])


def item_owner(items: np.ndarray, world: int):
    """(owner rank, local row at the owner) of global item ids under the interleaved row sharding"""
    items = np.asarray(items)
    return items % world, items // world


def shard_item_rows(Q: np.ndarray, world: int, rank: int) -> np.ndarray:
    """the rows of a [n_items, d] table that live on ``rank`` (items rank, rank+world, ...), in local row order"""
    return np.ascontiguousarray(Q[rank::world])


# ---- control plane ------------------------------------------------------------------------------------------------
class ControlPlane:
    """Host-side rendezvous of the ranks over torch.distributed/gloo (env MASTER_ADDR, MASTER_PORT, RANK, WORLD_SIZE as
    set by ``torch.distributed.run``).  Small host values only."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # every collective is preceded by an exchange of (operation, call site) tags (``_agree``): ranks that have stopped calling the SAME
        # collective -- one in an all-gather, the other already at a barrier -- raise at once, naming both sites, instead of waiting for each
        # other until the transport's timeout (round 5: a leg of bench.py behind a condition only rank 0 satisfied hung the N > 1 line)
        self.check = os.environ.get("QREC_CONTROL_CHECK", "1") != "0" and self.world > 1
        self._warned_sites = False

    @classmethod
    def from_env(cls):
        import torch.distributed as dist
        if not dist.is_initialized():
            import datetime
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29571")
            # a collective the other ranks never join fails after QREC_CONTROL_TIMEOUT seconds.  Default = gloo's own half hour (round 5's
            # 900 s could kill a healthy job whose rank 0 spends longer than that between two collectives -- an fp64 reference run at 25 M
            # triplets, an evaluation on a large test set); ranks in DIFFERENT collectives are caught at once by the tag exchange (_agree), a
            # dead communicator by the preflight's watchdog -- the transport's timeout is only the last resort
            dist.init_process_group("gloo", rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")),
                                    timeout=datetime.timedelta(seconds=float(os.environ.get("QREC_CONTROL_TIMEOUT", "1800"))))
        return cls()

    def _agree(self, op: str):
        """all ranks are about to run the same collective: exchange crc32(op), crc32(call site); a different OPERATION on some rank raises
        RuntimeError on every rank with each rank's operation and site; the same operation from different sites only warns (once)."""
        if not self.check:
            return
        import sys as _sys
        import zlib
        f = _sys._getframe(2)
        here = os.path.abspath(__file__)
        while f.f_back is not None and os.path.abspath(f.f_code.co_filename) == here:      # the caller outside this module
            f = f.f_back
        site = f"{os.path.basename(f.f_code.co_filename)}:{f.f_code.co_name}:{f.f_lineno}"
        t = self._torch.tensor([zlib.crc32(op.encode()), zlib.crc32(site.encode())], dtype=self._torch.int64)
        tags = [self._torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(tags, t, group=self.group)
        ops, sites = {int(x[0]) for x in tags}, {int(x[1]) for x in tags}
        if len(ops) == 1 and (len(sites) == 1 or self._warned_sites):
            return
        text = f"{op} at {site}".encode()[:200]
        mine = self._torch.zeros(200, dtype=self._torch.uint8)
        mine[:len(text)] = self._torch.frombuffer(bytearray(text), dtype=self._torch.uint8)
        everyone = [self._torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(everyone, mine, group=self.group)
        where = "; ".join(f"rank {r}: {bytes(x.numpy().tobytes()).rstrip(bytes(1)).decode(errors='replace')}" for r, x in enumerate(everyone))
        if len(ops) > 1:
            raise RuntimeError(f"control-plane collectives diverged -- the ranks are not in the same collective: {where}")
        self._warned_sites = True
        print(f"qrec control plane (rank {self.rank}): the same collective from different call sites: {where}", file=_sys.stderr, flush=True)

    def barrier(self):
        self._agree("barrier")
        self._dist.barrier(group=self.group)

    def broadcast_bytes(self, payload: bytes | None, n: int, src: int = 0) -> bytes:
        self._agree("broadcast_bytes")
        t = self._torch.zeros(n, dtype=self._torch.uint8)
        if self.rank == src:
            t.copy_(self._torch.frombuffer(bytearray(payload), dtype=self._torch.uint8))
        self._dist.broadcast(t, src=src, group=self.group)
        return bytes(t.numpy().tobytes())

    def allreduce_host(self, arr: np.ndarray, op: str = "sum") -> np.ndarray:
        self._agree("allreduce_host")
        t = self._torch.from_numpy(np.array(arr, copy=True))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX if op == "max" else self._dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def allgather_host(self, arr: np.ndarray) -> np.ndarray:
        """[world, *arr.shape]; every rank passes the same shape"""
        self._agree("allgather_host")
        t = self._torch.from_numpy(np.ascontiguousarray(arr))
        out = [self._torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t, group=self.group)
        return np.stack([o.numpy() for o in out])

    def shutdown(self):
        if self._dist.is_initialized():
            self._agree("shutdown")
            self._dist.barrier(group=self.group)
            self._dist.destroy_process_group()


class GlooStagedComm:
    """FUNCTIONAL-TEST transport with ``capi.Comm``'s interface: device buffers are staged through the host and the
    control plane's gloo group.  Exists because RCCL refuses two ranks on one device and the development boxes have one
    GPU (QREC_DIST_TEST_ONE_DEVICE=1 puts every rank on device 0) -- it lets the N > 1 control flow, kernels and
    exchange bookkeeping run end to end on real hardware.  Synchronous; numbers from such runs are not bench results."""

    _NP = {_capi.F32: np.float32, _capi.F64: np.float64, _capi.I32: np.int32}

    def __init__(self, control: ControlPlane, kern=_capi):
        self.cp, self.k = control, kern
        self.world, self.rank = control.world, control.rank

    def _get(self, ptr, count, dtype, stream):
        out = np.empty(count, dtype=dtype)
        if count:
            self.k.memcpy_d2h(out, ptr, out.nbytes, stream)
        return out

    def _put(self, ptr, arr, stream):
        if arr.size:
            self.k.memcpy_h2d(ptr, np.ascontiguousarray(arr), arr.nbytes, stream)

    def allreduce(self, buf, count, dtype=_capi.F32, stream=None):
        self._put(buf, self.cp.allreduce_host(self._get(buf, count, self._NP[dtype], stream)), stream)

    def allreduce_pair(self, a, count_a, dtype_a, b, count_b, dtype_b, stream=None):
        self.allreduce(a, count_a, dtype_a, stream); self.allreduce(b, count_b, dtype_b, stream)

    def allgather(self, send, recv, count, dtype=_capi.F32, stream=None):
        self._put(recv, self.cp.allgather_host(self._get(send, count, self._NP[dtype], stream)).ravel(), stream)

    def reduce_scatter(self, send, recv, count, dtype=_capi.F32, stream=None):
        full = self.cp.allreduce_host(self._get(send, count * self.world, self._NP[dtype], stream))
        self._put(recv, full[self.rank * count:(self.rank + 1) * count], stream)

    def alltoall_rows(self, send, send_rows, recv, recv_rows, row_bytes, stream=None):
        torch, dist = self.cp._torch, self.cp._dist
        s = [int(x) * row_bytes for x in send_rows]; r = [int(x) * row_bytes for x in recv_rows]
        src = torch.from_numpy(self._get(send, sum(s), np.uint8, stream)); dst = torch.empty(sum(r), dtype=torch.uint8)
        self.cp._agree("alltoall_rows")
        dist.all_to_all_single(dst, src, r, s, group=self.cp.group)
        self._put(recv, dst.numpy(), stream)

    def sendrecv_segments(self, send, sends, recv, recvs, stream=None):
        """(peer, byte offset, bytes) lists, matched per pair in list order: one all_to_all_single of the per-peer concatenations"""
        torch, dist = self.cp._torch, self.cp._dist
        base_s, base_r = self.k.device_ptr(send) if send is not None else 0, self.k.device_ptr(recv) if recv is not None else 0
        out, s_sizes = [], []
        for p in range(self.world):
            seg = [self._get(base_s + o, nb, np.uint8, stream) for q, o, nb in sends if q == p and nb]
            out += seg; s_sizes.append(int(sum(x.size for x in seg)))
        r_sizes = [int(sum(nb for q, _, nb in recvs if q == p)) for p in range(self.world)]
        src = torch.from_numpy(np.concatenate(out) if out else np.zeros(0, np.uint8)); dst = torch.empty(sum(r_sizes), dtype=torch.uint8)
        self.cp._agree("sendrecv_segments")
        dist.all_to_all_single(dst, src, r_sizes, s_sizes, group=self.cp.group)
        got, at = dst.numpy(), 0
        for p in range(self.world):
            for q, o, nb in recvs:
                if q == p and nb:
                    self._put(base_r + o, got[at:at + nb], stream); at += nb

    def destroy(self):
        pass


def make_comm(control: ControlPlane):
    """the data-plane communicator of this process: RCCL (rank 0's id travels over the control plane); the staged gloo
    transport only under QREC_DIST_TEST_ONE_DEVICE=1"""
    if os.environ.get("QREC_DIST_TEST_ONE_DEVICE") == "1":
        return GlooStagedComm(control)
    uid = control.broadcast_bytes(_capi.comm_unique_id() if control.rank == 0 else None, _capi.COMM_UID_BYTES)
    return _capi.Comm(control.world, control.rank, uid)


class watchdog:
    """``with watchdog("what", seconds):`` -- if the block has not finished after ``seconds``, one line on stderr and ``os._exit(3)``.  For the
    calls of a multi-process start-up that block inside a library when the ranks do not all arrive (``ncclCommInitRank``): a hung job says
    nothing, a dead one says why.  ``on_hang(reason)`` replaces the exit (tests)."""

    def __init__(self, what: str, seconds: float, on_hang=None):
        self.what, self.seconds, self.on_hang, self._timer = what, float(seconds), on_hang, None

    def _fire(self):
        reason = f"qrec watchdog: {self.what} did not complete within {self.seconds:.0f} s"
        if self.on_hang is not None:
            self.on_hang(reason)
            return
        import sys as _sys
        print(reason, file=_sys.stderr, flush=True)
        os._exit(3)

    def __enter__(self):
        import threading
        self._timer = threading.Timer(self.seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


def preflight(comm, kern=_capi, stream=None, timeout_s: float = 90.0, on_hang=None) -> dict:
    """First contact with the data-plane communicator, before anything is timed (round 5): a tiny all-reduce and an
    ``alltoall_rows`` round trip with ragged row counts, both checked against what they must return, under a watchdog.

    * all-reduce: 256 floats, element k = rank + 1 + k  ->  world (world + 1) / 2 + world k on every rank;
    * all-to-all: rank r sends peer p  1 + (r + p) % 3  rows of 64 floats, every float = 1000 r + p + 1; what arrives from p must be
      1000 p + r + 1; the rows then travel BACK and must equal what was sent.
    A wrong value raises RuntimeError naming the collective, the rank and the first bad element.  A collective that never completes
    (mismatched ranks, an IPC mode the driver refuses, a dead peer) would otherwise hang the job silently: the checks run in a
    daemon thread, and when it has not finished after ``timeout_s`` seconds ``on_hang(reason)`` is called -- by default one line on
    stderr and ``os._exit(3)`` (a blocked HIP stream synchronise cannot be interrupted from Python).  Returns what was exchanged."""
    import threading
    world, rank = int(comm.world), int(comm.rank)
    result: dict = {}
    # HIP's current device is a property of the HOST THREAD: the checks run in a thread of their own, which would otherwise talk to device 0
    # whatever device this rank selected
    device = kern.current_device() if hasattr(kern, "current_device") else None

    def checks():
        try:
            if device is not None:
                kern.init(device)
            n = 256
            mine = (np.arange(n, dtype=np.float32) + (rank + 1)).astype(np.float32)
            buf = kern.DeviceBuffer.from_numpy(mine)
            comm.allreduce(buf, n, kern.F32, stream)
            got = buf.numpy(stream)
            want = (world * (world + 1) / 2 + world * np.arange(n)).astype(np.float32)
            if not np.array_equal(got, want):
                k = int(np.flatnonzero(got != want)[0])
                raise RuntimeError(f"preflight all-reduce: rank {rank} of {world} got {got[k]!r} at element {k}, expected {want[k]!r}")
            cols = 64
            s_rows = [1 + (rank + p) % 3 for p in range(world)]
            r_rows = list(s_rows)                                             # 1 + (p + r) % 3 is symmetric in (r, p)
            send = np.concatenate([np.full((s_rows[p], cols), 1000 * rank + p + 1, np.float32) for p in range(world)])
            want_in = np.concatenate([np.full((r_rows[p], cols), 1000 * p + rank + 1, np.float32) for p in range(world)])
            d_send, d_recv, d_back = kern.DeviceBuffer.from_numpy(send), kern.DeviceBuffer(want_in.shape, np.float32), kern.DeviceBuffer(send.shape, np.float32)
            comm.alltoall_rows(d_send, s_rows, d_recv, r_rows, cols * 4, stream)
            got_in = d_recv.numpy(stream)
            if not np.array_equal(got_in, want_in):
                k = int(np.flatnonzero((got_in != want_in).any(1))[0])
                raise RuntimeError(f"preflight all-to-all: rank {rank} of {world} received {got_in[k, 0]!r} in row {k}, expected {want_in[k, 0]!r}")
            comm.alltoall_rows(d_recv, r_rows, d_back, s_rows, cols * 4, stream)
            back = d_back.numpy(stream)
            if not np.array_equal(back, send):
                k = int(np.flatnonzero((back != send).any(1))[0])
                raise RuntimeError(f"preflight all-to-all round trip: rank {rank} of {world} got row {k} back as {back[k, 0]!r}, sent {send[k, 0]!r}")
            result.update(ok=True, world=world, allreduce_floats=n, alltoall_rows_sent=int(sum(s_rows)), row_bytes=cols * 4,
                          checksum=float(np.float64(got.sum()) + np.float64(got_in.sum())))
        except BaseException as e:      # noqa: BLE001 -- handed to the caller's thread
            result["error"] = e

    th = threading.Thread(target=checks, name="qrec-preflight", daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        reason = (f"qrec preflight: rank {rank} of {world}: the first collectives did not complete within {timeout_s:.0f} s "
                  f"(communicator {type(comm).__name__}; check that all {world} ranks started, HSA_ENABLE_IPC_MODE_LEGACY=0, one GPU per rank)")
        if on_hang is not None:
            on_hang(reason)
            return {"ok": False, "reason": reason}
        import sys as _sys
        print(reason, file=_sys.stderr, flush=True)
        os._exit(3)
    if "error" in result:
        raise result["error"]
    return result


# ---- BPR, replicated item table ------------------------------------------------------------------------------------
class ReplicatedTableSync:
    """Delta all-reduce of a replicated fp32 table resident on the device.  ``sync()`` after a step makes every replica
    equal to start + sum of all ranks' deltas and re-arms the snapshot: three enqueues on the caller's stream (delta
    kernel, all-reduce, apply kernel), nothing on the host.  ``extra`` = (buffer, count, dtype) rides in the same fused
    collective launch (the epoch's loss terms)."""

    def __init__(self, comm, table, n_floats: int | None = None, kern=_capi, stream=None):
        self.comm, self.k, self.table = comm, kern, table
        self.n = int(n_floats if n_floats is not None else table.nbytes // 4)
        self.start = kern.DeviceBuffer(self.n, np.float32)
        self.delta = kern.DeviceBuffer(self.n, np.float32)
        kern.memcpy_d2d(self.start, table, self.n * 4, stream)

    def sync(self, stream=None, extra=None):
        k = self.k
        k.table_delta(self.table, self.start, self.delta, self.n, stream)
        if extra is None:
            self.comm.allreduce(self.delta, self.n, k.F32, stream)
        else:
            self.comm.allreduce_pair(self.delta, self.n, k.F32, extra[0], extra[1], extra[2], stream)
        k.table_apply(self.table, self.start, self.delta, self.n, stream)


# ---- BPR, row-sharded item table -----------------------------------------------------------------------------------
class ShardedItemExchange:
    """Cross-shard row lookups of one rank for an epoch of BPR triplets (module docstring, ``sharded``).

    ``plan_epoch`` runs on the device for ALL batches in one set of launches (distinct rows per owner, triplet ids rewritten
    to cache slots), costs ONE host synchronisation per epoch -- the row counts, which size the exchanges -- and ships the
    request ids of every batch to their owners in one fused launch (``sendrecv_segments``).  ``run_epoch`` then only enqueues,
    per batch: owners gather the requested rows, rows travel to the requesters' cache, ``sgd_batch`` trains on the cache,
    the cache travels back, owners add ``returned - sent`` into their rows.
    Every rank must call both with the same ``n_batches`` (ranks hold different triplet counts; a rank that has run out of
    triplets still serves its rows).  ``kern`` is the C ABI binding (tests substitute a host emulation to check the protocol
    on CPU over gloo).

    ``pipeline`` = (second communicator, fetch stream), round 3 ("overlap step k+1 fetch with step k compute"):
    the fetch of batch k + 1 (gather + row exchange into the OTHER cache buffer) is enqueued on the fetch stream and runs
    under batch k's SGD kernel; ordered by events so that it is deterministic -- the gather of batch k + 1 happens after the
    owners applied batch k - 1 and before they apply batch k: a batch sees the item table as of TWO batches back (one more
    batch of staleness than the unpipelined protocol, the same kind the replicated layout has per epoch).  Nothing is lost:
    owners still add ``returned - sent`` of every batch.  The second communicator exists because two collectives of one
    communicator must not be in flight on two streams."""

    def __init__(self, comm, n_items: int, ld: int, q_local, kern=_capi, pipeline=None, plan_ahead=None):
        self.comm, self.k = comm, kern
        self.world, self.rank = comm.world, comm.rank
        self.n_items, self.ld, self.q_local = int(n_items), int(ld), q_local
        self.rows_local = kern.shard_rows(self.n_items, self.world, self.rank)
        self._cap = {}
        self._scratch_batches = 0
        self._uploaded = {}
        self.n = self.n_batches = 0
        self.bounds = self.send = self.recv = self.req_off = self.in_off = None
        self.bytes_moved = 0
        self.pipeline = pipeline
        if pipeline is not None:
            self.comm_f, self.stream_f = pipeline
            ev = kern.Event
            self.ev_fetched, self.ev_gathered, self.ev_free = [ev(), ev()], [ev(), ev()], [ev(), ev()]
            self.ev_epoch_start = ev()
        # ``plan_ahead`` = (third communicator, plan stream), round 3: the plan of epoch k + 1 -- it depends on nothing but that
        # epoch's negatives, which the sampler draws under epoch k -- runs on the plan stream under epoch k, its one host read-back
        # and its id exchange included; the training stream only waits for an event.  Plan buffers alternate between two slots.
        self.plan_ahead = plan_ahead
        self._slot, self._ahead = 0, None
        self._pinned, self._counted = {}, {}
        if plan_ahead is not None:
            self.comm_p, self.stream_p = plan_ahead
            self.ev_planned = [kern.Event(), kern.Event()]
            self.ev_plan_free = [None, None]        # recorded on the training stream after the last reader of a slot's id arrays

    def _buf(self, name: str, elems: int, dtype):
        """grow-only device buffer"""
        b = self._cap.get(name)
        if b is None or b.nbytes < elems * np.dtype(dtype).itemsize:
            b = self._cap[name] = self.k.DeviceBuffer(max(int(elems * 1.25), 1), dtype)
        return b

    def _const(self, name: str, values: np.ndarray, stream):
        """small host array kept on the device; uploaded again only when its contents change (qrec_memcpy_h2d returns after
        the copy, i.e. it drains the stream: two of them per epoch kept the host from ever running ahead of the device)"""
        values = np.ascontiguousarray(values)
        d = self._buf(name, values.size, values.dtype)
        old = self._uploaded.get(name)
        if old is None or old[0] is not d or not np.array_equal(old[1], values):
            self.k.memcpy_h2d(d, values, values.nbytes, stream)        # (returns after the copy: the host array is borrowed)
            self._uploaded[name] = (d, values.copy())
        return d

    def _plan_begin(self, d_i, d_j, n: int, n_batches: int, bounds, comm, stream, slot: int):
        """the device part of a plan, enqueued on (comm, stream) into the buffers of ``slot``: distinct rows per owner and batch,
        triplet ids rewritten to cache slots, the counts of every rank gathered and on their way to page-locked host memory,
        an event behind that copy.  Returns the plan's description so far."""
        k, G = self.k, self.world
        n, nb = int(n), int(n_batches)
        per = -(-n // nb) if n else 0
        bounds = [int(x) for x in bounds] if bounds is not None else [min(b * per, n) for b in range(nb + 1)]
        if len(bounds) != nb + 1 or bounds[0] != 0 or bounds[-1] != n:
            raise ValueError("plan_epoch: bounds must run from 0 to n in n_batches steps")
        caps = [min(2 * (bounds[b + 1] - bounds[b]), self.n_items) for b in range(nb)]
        req_off = np.concatenate([[0], np.cumsum(caps)]).astype(np.int64)
        sfx = f"@{slot}"
        d_req = self._buf("req" + sfx, int(req_off[-1]), np.int32)
        d_ci, d_cj = self._buf("ci" + sfx, n, np.int32), self._buf("cj" + sfx, n, np.int32)
        d_counts = self._buf("counts" + sfx, nb * G, np.int32)
        d_all = self._buf("all_counts" + sfx, nb * G * G, np.int32)
        if self._scratch_batches < nb:
            self._cap["scratch"] = k.DeviceBuffer(k.shard_plan_epoch_scratch_bytes(self.n_items, G, nb), np.uint8)
            self._scratch_batches = nb
        d_bounds = self._const("bounds" + sfx, np.array(bounds, np.int64), stream)      # per plan slot: a re-upload for one slot (plan stream) must
        d_roff = self._const("req_off" + sfx, req_off, stream)                          # not race a plan kernel of the other (training stream)
        k.shard_plan_epoch(d_i, d_j, d_bounds, nb, n, self.n_items, G, self._cap["scratch"], d_req, d_roff, d_counts, d_ci, d_cj, stream)
        comm.allgather(d_counts, d_all, nb * G, k.I32, stream)
        pin = self._pinned.get(slot)
        if pin is None or pin.nbytes < 4 * nb * G * G:
            pin = self._pinned[slot] = k.PinnedBuffer(nb * G * G, np.int32)
            self._counted[slot] = k.Event()
        k.memcpy_d2h_async(pin, d_all, 4 * nb * G * G, stream)
        self._counted[slot].record(stream)
        return dict(n=n, n_batches=nb, bounds=bounds, req_off=req_off, slot=slot, d_j=d_j, finished=False)

    def _plan_finish(self, plan, comm, stream):
        """the host part: wait for the counts (the plan's one host synchronisation -- of the copy's event, not of the stream),
        size the exchanges, ship the request ids of ALL batches to their owners in one fused launch on (comm, stream)"""
        k, G, nb, slot = self.k, self.world, plan["n_batches"], plan["slot"]
        self._counted[slot].sync()
        counts = self._pinned[slot].a[:G * nb * G].reshape(G, nb, G)
        send = counts[self.rank].astype(np.int64)                        # [batch][owner]: rows I ask of each owner
        recv = counts[:, :, self.rank].T.astype(np.int64).copy()         # [batch][peer]:  rows each peer asks of me
        in_off = np.concatenate([[0], np.cumsum(recv.sum(1))]).astype(np.int64)
        d_req, d_req_in = self._cap[f"req@{slot}"], self._buf(f"req_in@{slot}", int(in_off[-1]), np.int32)
        sends, recvs = [], []       # for every (batch, peer) one send and one receive, both sides in batch order
        for b in range(nb):
            so, ro = int(plan["req_off"][b]), int(in_off[b])
            for p in range(G):
                sends.append((p, 4 * so, 4 * int(send[b, p]))); so += int(send[b, p])
                recvs.append((p, 4 * ro, 4 * int(recv[b, p]))); ro += int(recv[b, p])
        comm.sendrecv_segments(d_req, sends, d_req_in, recvs, stream)
        plan.update(send=send, recv=recv, in_off=in_off, finished=True)
        return plan

    def _plan(self, d_i, d_j, n: int, n_batches: int, bounds, comm, stream, slot: int):
        return self._plan_finish(self._plan_begin(d_i, d_j, n, n_batches, bounds, comm, stream, slot), comm, stream)

    def _adopt(self, plan):
        """make ``plan`` the epoch ``run_epoch`` runs; size the row buffers for it"""
        self.n, self.n_batches, self.bounds = plan["n"], plan["n_batches"], plan["bounds"]
        self.req_off, self.send, self.recv, self.in_off = plan["req_off"], plan["send"], plan["recv"], plan["in_off"]
        self._slot = plan["slot"]
        r_in, r_out = int(self.recv.sum(1).max(initial=0)), int(self.send.sum(1).max(initial=0))
        copies = 2 if self.pipeline is not None else 1
        for c in range(copies):
            self._buf(f"rows_out{c}", r_in * self.ld, np.float32); self._buf(f"cache{c}", r_out * self.ld, np.float32)
        self._buf("rows_ret", r_in * self.ld, np.float32)

    def plan_epoch(self, d_i, d_j, n: int, n_batches: int, stream=None, bounds=None):
        """``bounds``: the batches' triplet ranges (n_batches + 1 offsets); default = equal consecutive ranges.
        With a plan made ahead for exactly these negatives (``plan_epoch_ahead``) this only makes ``stream`` wait for it."""
        ahead, self._ahead = self._ahead, None
        if ahead is not None and ahead["d_j"] is d_j and ahead["n"] == int(n) and ahead["n_batches"] == int(n_batches):
            if ahead["finished"]:        # made on the plan stream (plan_epoch_ahead)
                self.k.stream_wait_event(stream, self.ev_planned[ahead["slot"]])
            else:                        # begun inside the previous epoch, on this stream (run_epoch(next_epoch=...)): its counts are on
                self._plan_finish(ahead, self.comm, stream)       # the host long before the stream runs dry
            self._adopt(ahead)
            return
        if ahead is not None:            # a plan for other negatives: its stream work must not outlive its buffers' next use
            (self.ev_planned[ahead["slot"]] if ahead["finished"] else self._counted[ahead["slot"]]).sync()
        # the slot that no plan in flight can be using
        self._adopt(self._plan(d_i, d_j, n, n_batches, bounds, self.comm, stream, self._slot ^ 1))

    def plan_epoch_ahead(self, d_i, d_j_next, n: int, n_batches: int, after, bounds=None):
        """Plan the NEXT epoch from its negatives ``d_j_next`` on the plan stream: ``after`` = the event recorded behind the sampler
        that draws them.  Blocks the host until that plan's row counts are known -- the device meanwhile runs the current epoch --
        and leaves the plan for the ``plan_epoch`` call of the next epoch."""
        if self.plan_ahead is None:
            raise RuntimeError("ShardedItemExchange was built without plan_ahead=(communicator, stream)")
        slot = self._slot ^ 1
        P = self.stream_p
        self.k.stream_wait_event(P, after)
        if self.ev_plan_free[slot] is not None:      # the epoch that last trained from this slot's id arrays
            self.k.stream_wait_event(P, self.ev_plan_free[slot])
        self._ahead = self._plan(d_i, d_j_next, n, n_batches, bounds, self.comm_p, P, slot)
        self.ev_planned[slot].record(P)

    def _fetch(self, b, comm, stream):
        """owners answer batch b: gather the requested rows, ship them to the requesters' cache (buffer b % copies)"""
        k, c, ld = self.k, self._cap, self.ld
        cp = b & 1 if self.pipeline is not None else 0
        n_in = int(self.recv[b].sum())
        k.gather_rows(self.q_local, ld, k.device_ptr(c[f"req_in@{self._slot}"]) + 4 * int(self.in_off[b]), n_in, c[f"rows_out{cp}"], stream)
        if self.pipeline is not None:
            self.ev_gathered[cp].record(stream)
        comm.alltoall_rows(c[f"rows_out{cp}"], self.recv[b], c[f"cache{cp}"], self.send[b], ld * 4, stream)
        return cp

    def run_epoch(self, sgd_batch, stream=None, next_epoch=None, before_first_sgd=None):
        """``sgd_batch(t0, n, d_cache, cache_rows, ci_ptr, cj_ptr, stream)`` trains triplets [t0, t0+n) whose item ids
        are rows of ``d_cache`` (addresses of the rewritten id arrays are passed).

        ``next_epoch`` = dict(d_i, d_j, n, n_batches, bounds, after) -- the NEXT epoch's triplets, ``after`` the event behind the
        sampler that draws its negatives (round 3): the device part of that epoch's plan is enqueued on ``stream`` in front of
        this epoch's LAST batch, so that its row counts reach the host while that batch still trains and the next
        ``plan_epoch`` finds them there -- the host never waits for the stream to run dry, and no plan kernel runs beside an SGD
        grid (which costs the atomic-bound grid more than it hides: profiles/r03_sharded_world1.json).  Epochs of one batch
        have no "in front of the last batch" that the sampler could be ready for; they plan at their own start.
        ``before_first_sgd(stream)``: called where the first batch's SGD grid goes, behind its fetch."""
        k, c, ld = self.k, self._cap, self.ld
        ci, cj = k.device_ptr(c[f"ci@{self._slot}"]), k.device_ptr(c[f"cj@{self._slot}"])
        piped = self.pipeline is not None
        if piped:
            F = self.stream_f
            # everything enqueued on `stream` so far -- the plan, the id exchange, the previous epoch's last batch -- comes first
            self.ev_epoch_start.record(stream); k.stream_wait_event(F, self.ev_epoch_start)
            self._fetch(0, self.comm_f, F); self.ev_fetched[0].record(F)
        for b in range(self.n_batches):
            t0, nb = self.bounds[b], self.bounds[b + 1] - self.bounds[b]
            S, R = self.send[b], self.recv[b]
            n_out, n_in = int(S.sum()), int(R.sum())
            ne = None
            if next_epoch is not None and b == self.n_batches - 1 and b >= 1:
                ne = next_epoch() if callable(next_epoch) else next_epoch     # (callable: asked only now -- the sampler it waits for was enqueued by the first batch)
            if ne is not None:
                k.stream_wait_event(stream, ne["after"])
                self._ahead = self._plan_begin(ne["d_i"], ne["d_j"], ne["n"], ne["n_batches"], ne.get("bounds"), self.comm, stream, self._slot ^ 1)
            if piped:
                cp = b & 1
                if b + 1 < self.n_batches:      # the next batch's fetch goes out now, into the other buffer (free once batch b - 1 was applied)
                    if b >= 1:
                        k.stream_wait_event(F, self.ev_free[(b + 1) & 1])
                    self._fetch(b + 1, self.comm_f, F); self.ev_fetched[(b + 1) & 1].record(F)
                k.stream_wait_event(stream, self.ev_fetched[cp])
            else:
                cp = self._fetch(b, self.comm, stream)
            cache, rows_out = c[f"cache{cp}"], c[f"rows_out{cp}"]
            if b == 0 and before_first_sgd is not None:
                before_first_sgd(stream)         # on every rank, with or without triplets in this batch: what it enqueues (the next epoch's
                                                 # sampler) decides whether the rank joins the plan's collectives in front of the last batch
            if nb:
                sgd_batch(t0, nb, cache, n_out, ci + 4 * t0, cj + 4 * t0, stream)
            self.comm.alltoall_rows(cache, S, c["rows_ret"], R, ld * 4, stream)                          # rows -> owners
            if piped and b + 1 < self.n_batches:
                k.stream_wait_event(stream, self.ev_gathered[(b + 1) & 1])     # batch b + 1 was gathered BEFORE batch b is applied
            k.scatter_add_row_deltas(self.q_local, ld, k.device_ptr(c[f"req_in@{self._slot}"]) + 4 * int(self.in_off[b]), n_in, c["rows_ret"], rows_out, stream)
            if piped:
                self.ev_free[cp].record(stream)
            off = n_out - int(S[self.rank])
            self.bytes_moved += off * (4 + 2 * ld * 4)        # what left this rank for other ranks
        if self.plan_ahead is not None:       # this slot's id arrays are free for the plan after next once the stream gets here
            if self.ev_plan_free[self._slot] is None:
                self.ev_plan_free[self._slot] = k.Event()
            self.ev_plan_free[self._slot].record(stream)


class ReplicatedStep:
    """what ``BprSgd.epoch_device_async(dist=...)`` needs for the replicated layout: the item table's sync and, when the
    user table is replicated too (drop-in classes), its sync"""
    mode = "replicated"

    def __init__(self, comm, sync_q: ReplicatedTableSync, sync_p: ReplicatedTableSync | None = None):
        self.comm, self.sync_q, self.sync_p = comm, sync_q, sync_p

    def sync_tables(self, stream=None):
        """reconcile the replicas INSIDE an epoch (``BprSgd(batches=K)``: after each of the first K - 1 batches; the last batch's
        sync is the epoch close's fused collective)"""
        # (round 4 could restrict the inner reconciliations to the rows with the most positives; round 5 removed it: 4 % of the epoch at 8
        # ranks for a Recall gap twice the whole table's -- profiles/r05_strong_scaling_bound.json, r05_scaling_recall.json)
        self.sync_q.sync(stream)
        # (the user table, when it is replicated too -- the drop-in classes -- is NOT reconciled here: users are sharded over the ranks,
        # no two ranks touch the same row of P, so its one delta-sum at the epoch close is exact whenever it happens)


class ShardedStep:
    """... and for the row-sharded layout: the exchange and the number of batches per epoch every rank agreed on.
    ``prepare`` (after the epoch's negatives are on the device, before ``epoch_device_async``) plans the epoch."""
    mode = "sharded"

    def __init__(self, comm, exchange: ShardedItemExchange, n_batches: int, plan_inside: bool = True):
        self.comm, self.exchange, self.n_batches, self.plan_inside = comm, exchange, int(n_batches), bool(plan_inside)

    def prepare(self, sgd, stream=None):
        self.exchange.plan_epoch(sgd.d_i, sgd.d_j, sgd.n, self.n_batches, stream, bounds=sgd.batch_bounds)

    def prepare_ahead(self, sgd):
        """after ``sgd.prefetch_negatives_device``: plan the next epoch from the negatives being drawn (no-op without plan_ahead)"""
        if self.exchange.plan_ahead is not None:
            self.exchange.plan_epoch_ahead(sgd.d_i, sgd.d_j_next, sgd.n, self.n_batches, sgd._sampled, bounds=sgd.batch_bounds)

    def next_epoch(self, sgd):
        """what ``run_epoch(next_epoch=...)`` needs, when the next epoch's negatives are being drawn already and no plan stream is in use"""
        if self.exchange.plan_ahead is not None or getattr(sgd, "_prefetched_epoch", None) is None or not self.plan_inside:
            return None
        return dict(d_i=sgd.d_i, d_j=sgd.d_j_next, n=sgd.n, n_batches=self.n_batches, bounds=sgd.batch_bounds, after=sgd._sampled)


def agree_on_batches(control: ControlPlane, n_local: int, batch: int, split_from: int = 0, min_batches: int = 0) -> int:
    """batches per epoch such that no rank's batch exceeds ``batch`` triplets (the rank with the most triplets decides);
    ``split_from`` > 0: at least two batches once that rank holds ``split_from`` triplets -- with two or more batches the next
    epoch's plan hides in front of the last one (``ShardedItemExchange.run_epoch``); ``min_batches``: at least that many
    (``reconciliations_per_epoch``)"""
    n_max = int(control.allreduce_host(np.array([n_local], dtype=np.int64), op="max")[0])
    nb = max(1, -(-n_max // max(int(batch), 1)), int(min_batches))
    return max(nb, 2) if split_from and n_max >= split_from else nb


def reconciliations_per_epoch(world: int, requested: int = 0) -> int:
    """How many times per epoch the ranks' copies of the item rows are reconciled (replicated layout: delta all-reduces; sharded
    layout: at least that many exchange batches).  0 = the default: 1 up to two ranks, 2 beyond.

    Between two reconciliations a rank does not see what the others did to the rows they share: of the T updates an item row takes per
    epoch, a rank misses (G - 1) / G of those in its window of 1 / K epoch -- a share (G - 1) / (G K) of T.  Measured with the paired
    Recall@20 design on the planted-community graph at BPR.conf's rate (tools/paired_recall.py; |Recall@20 - order-exact training of
    the whole problem| at the reference's peak epoch / at the last epoch, two seeds where two numbers are given):

        G  K   missed share   peak            final            epoch (one rank's share, links excluded)
        2  1   0.50           0.0009 0.0008   0.0011 0.0007    0.373 ms  1.71x
        4  1   0.75           0.0029          0.0016           0.224 ms  2.84x      <- outside the +-0.002 bar
        4  2   0.38           0.0013 0.0010   0.0007 0.0012    0.252 ms  2.53x
        4  4   0.19           0.0006          0.0000           0.288 ms  2.21x
        8  1   0.88           0.0016          0.0031           0.158 ms  4.02x      <- outside
        8  2   0.44           0.0007 0.0008   0.0007 0.0009    0.178 ms  3.58x
        8  4   0.22           0.0003 0.0004   0.0004 0.0003    0.218 ms  2.92x
        8  8   0.11           0.0002          0.0000           0.310 ms  2.05x

    (profiles/r04_paired_recall_studies.json part B, profiles/r05_scaling_recall.json, profiles/r05_strong_scaling_bound.json).  Every
    setting with a missed share of 0.5 or less is inside the bar at the peak AND at the last epoch; rounds 1-4 reconciled once per rank
    (K = G: the safest row of each block and the slowest -- eight whole-table delta / apply passes and eight launches of 19 k triplets
    at 8 ranks).  The default is now the smallest K with (G - 1) / (G K) <= 0.5."""
    if int(requested) > 0:
        return int(requested)
    return 1 if int(world) <= 2 else 2


# ---- graph models (LightGCN / NGCF / SimGCL ...) ---------------------------------------------------------------------
def _buffer_view(buf):
    """(device address, element count, capi dtype code) of a DeviceBuffer or a ``head_view``"""
    itf = getattr(buf, "__cuda_array_interface__", None)
    if itf is None:
        raise TypeError("all_reduce needs a DeviceBuffer or a DeviceBuffer.head_view")
    code = {"<f4": _capi.F32, "<f8": _capi.F64, "<i4": _capi.I32}[itf["typestr"]]
    return itf["data"][0], int(np.prod(itf["shape"], dtype=np.int64)), code


class BatchParallel:
    """SURVEY s8e, config #5.  Every rank holds the whole embedding table (17.8 MB at the Yelp2018 shape), the graph
    and the Adam slots, and propagates the whole graph; a training step covers ``batch_size x world`` consecutive
    rows of the reference's batch stream (base/deepRecommender.py:29-52, drawn identically on every rank from a
    broadcast seed), of which rank r takes the r-th contiguous share.  The pairwise loss is a sum over rows and the
    backward propagation is linear, so the sum over ranks of the dense table gradient (and of NGCF's four d x d weight
    gradients -- the north star's "all-reduce for the dense layers") is the gradient of the whole step: ONE all-reduce
    of N x ld floats per step (RCCL), then the same Adam update everywhere, which keeps the replicas bit-identical.
    A run on G ranks with ``batch_size = B`` therefore equals a single-GPU run with ``batch_size = G B`` up to fp32
    summation order.  Evaluation shards the test users; the per-user hit / DCG sums are disjoint and add exactly."""

    def __init__(self, control: ControlPlane, comm):
        self.control, self.comm = control, comm
        self.world, self.rank = control.world, control.rank

    @classmethod
    def from_env(cls):
        """the communicator ``init_from_env`` made, or None on one GPU"""
        return _STATE.get("dp")

    def share(self, n_rows: int) -> tuple[int, int]:
        """(offset, count) of this rank's contiguous share of a step's rows (counts differ by <= 1)"""
        lo, hi = user_block(n_rows, self.world, self.rank)
        return lo, hi - lo

    def all_reduce(self, buf, stream=None):
        """sum a device buffer over the ranks in place; ordered after the kernels enqueued so far on ``stream``"""
        ptr, count, code = _buffer_view(buf)
        self.comm.allreduce(ptr, count, code, stream)

    def all_reduce_host(self, arr: np.ndarray) -> np.ndarray:
        return self.control.allreduce_host(np.ascontiguousarray(arr))


class RowPartition:
    """1-D row partition of a propagation  Y = A X  over the ranks: rank r owns the contiguous row
    block [lo_r, hi_r) of A, of X and of Y, padded to ``rows_pad`` rows per rank so that the collectives are uniform.
    Forward: all-gather of the operand blocks, local SpMM of the rank's rows.  Backward (dX = A^T dY with only the
    rank's rows of dY): local product of the transposed row block into a full-height buffer, reduce-scatter."""

    def __init__(self, comm, n_rows: int, ld: int, kern=_capi):
        self.comm, self.k, self.n_rows, self.ld = comm, kern, int(n_rows), int(ld)
        self.world, self.rank = comm.world, comm.rank
        # rows per rank, rounded up to a multiple of 32: a block then starts on a word boundary of a row BITMAP over the whole table, so the
        # batch-row masks of the single-GPU steps (qrec_mark_batch_rows) address a block's rows by a pointer offset (round 4)
        self.rows_pad = (-(-self.n_rows // self.world) + 31) // 32 * 32
        self.lo = min(self.rank * self.rows_pad, self.n_rows)
        self.hi = min(self.lo + self.rows_pad, self.n_rows)

    def gather_operand(self, d_block, d_full, stream=None):
        """d_block [rows_pad][ld] (this rank's rows, pad rows zero) -> d_full [world*rows_pad][ld]"""
        self.comm.allgather(d_block, d_full, self.rows_pad * self.ld, self.k.F32, stream)

    def scatter_sum(self, d_full, d_block, stream=None):
        """d_full [world*rows_pad][ld] partial products of every rank -> d_block = this rank's rows of their sum"""
        self.comm.reduce_scatter(d_full, d_block, self.rows_pad * self.ld, self.k.F32, stream)

    # ---- referenced rows only (graph row: "or all-to-all of only referenced remote rows"), round 3 ---------------
    def reference(self, indptr: np.ndarray, indices: np.ndarray):
        """Prepare the exchange that ships, per product, only the operand rows this rank's block of the adjacency refers to.
        ``(indptr, indices)``: the WHOLE adjacency (every rank holds it when the trainer is built, so every rank can work out
        every other rank's needs without talking).  Returns the block's column indices renumbered into the compact operand
        [own rows_pad rows | rows needed from rank 0 | from rank 1 | ...] (ascending global row inside a peer's segment), in the
        block's stored order -- the SpMM then adds a row's terms in the same order as with global columns: same bits."""
        k, G, pad = self.k, self.world, self.rows_pad
        blocks = [(min(p * pad, self.n_rows), min(min(p * pad, self.n_rows) + pad, self.n_rows)) for p in range(G)]
        need = []
        for lo, hi in blocks:
            c = np.unique(indices[int(indptr[lo]):int(indptr[hi])])
            need.append(c[(c < lo) | (c >= hi)])
        mine = need[self.rank]
        owner = np.minimum(mine // pad, G - 1) if mine.size else np.zeros(0, np.int64)
        self.ref_recv_rows = np.bincount(owner, minlength=G).astype(np.int64)             # rows I receive from each peer
        sends = [need[p][(need[p] >= self.lo) & (need[p] < self.hi)] - self.lo for p in range(G)]      # my local rows each peer needs
        self.ref_send_rows = np.array([x.size for x in sends], np.int64)
        self.ref_send_ids = k.DeviceBuffer.from_numpy(np.concatenate(sends).astype(np.int32) if sum(x.size for x in sends) else np.zeros(1, np.int32))
        self.ref_send_buf = k.DeviceBuffer((max(int(self.ref_send_rows.sum()), 1), self.ld), np.float32)
        self.ref_rows = pad + int(mine.size)                                                 # height of the compact operand
        col_map = np.full(self.n_rows, -1, np.int64)
        col_map[self.lo:self.hi] = np.arange(self.hi - self.lo)
        col_map[mine] = pad + np.arange(mine.size)            # `mine` is ascending: grouped by owner, ascending inside a group
        blk = indices[int(indptr[self.lo]):int(indptr[self.hi])]
        self.ref_bytes_per_product = int(self.ref_send_rows.sum() - self.ref_send_rows[self.rank]) * self.ld * 4
        return col_map[blk].astype(np.int32)

    def gather_referenced(self, d_block, d_compact, stream=None):
        """d_block [rows_pad][ld] -> d_compact [ref_rows][ld]: own rows in front, behind them the remote rows the block refers to"""
        k, ld = self.k, self.ld
        k.memcpy_d2d(d_compact, d_block, self.rows_pad * ld * 4, stream)
        n_send = int(self.ref_send_rows.sum())
        if n_send:
            k.gather_rows(d_block, ld, self.ref_send_ids, n_send, self.ref_send_buf, stream)
        self.comm.alltoall_rows(self.ref_send_buf, self.ref_send_rows, k.device_ptr(d_compact) + self.rows_pad * ld * 4, self.ref_recv_rows,
                                ld * 4, stream)


_STATE: dict = {}


def init_from_env():
    """``python -m torch.distributed.run --nproc-per-node G -m qrec_amd.main <conf>``: one process per GPU.  Joins the
    control plane, binds the rank's device, creates the RCCL communicator and gives every rank the same ``random`` /
    ``numpy.random`` streams -- the reference never seeds, so rank 0's choice (QREC_SEED or the clock) is
    broadcast.  Returns the world size; 1 (and no torch import) when not launched that way."""
    import random
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1
    import time
    _capi.load()      # before torch: bind /opt/rocm's HIP runtime and its librccl, not the ones bundled with torch (comm.cpp)
    control = ControlPlane.from_env()
    one_device = os.environ.get("QREC_DIST_TEST_ONE_DEVICE") == "1"
    local = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["QREC_DEVICE"] = str(local)
    _capi.init(local)
    limit = float(os.environ.get("QREC_PREFLIGHT_TIMEOUT", "90"))
    with watchdog(f"rank {control.rank} of {world}: creating the RCCL communicator (ncclCommInitRank waits for ALL ranks)", limit):
        comm = make_comm(control)
    preflight(comm, timeout_s=limit)          # a tiny all-reduce and a ragged all-to-all round trip, checked, before anything trains
    seed = int(control.allreduce_host(np.array([int(os.environ.get("QREC_SEED", time.time_ns() % (2 ** 31)))
                                                if control.rank == 0 else 0], dtype=np.int64))[0])
    random.seed(seed); np.random.seed(seed % (2 ** 32))
    attach(control, comm)
    return world


def attach(control: ControlPlane, comm):
    """make (control, comm) this process's multi-GPU context: what ``BatchParallel.from_env`` / ``is_output_rank`` see"""
    _STATE.update(control=control, comm=comm, dp=BatchParallel(control, comm) if control.world > 1 else None)


def is_output_rank() -> bool:
    """result / measure files are written by rank 0 only"""
    control = _STATE.get("control")
    return control is None or control.rank == 0
