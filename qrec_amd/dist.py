"""Multi-GPU data parallelism of the BPR hot path -- one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the device, "gloo" in CPU tests).

Sharding: users are split into contiguous id blocks, one per rank.  A rank owns its users'
rows of ``P``, their triplets (the PositiveSet CSR is user-major, so this is a row split) and
their negative sampling: no exchange on that side.  The item table ``Q`` is replicated; every
rank applies its own triplets to its replica and, at the end of a step, the replicas are
reconciled by summing the per-rank deltas:

    Q  <-  Q_start + sum_r (Q_r - Q_start)            (one all-reduce of |Q| floats)

i.e. every rank's updates are kept (none is averaged away), and an item row read during a step
lags other ranks' updates by at most one step -- the same bounded-staleness contract the
single-GPU throughput kernel has inside a launch.  Q is 9.7 MB at the Yelp2018 shape and
0.5 GB at config #4: small next to 288 GB of HBM, and one ring all-reduce per step moves far
fewer bytes over the point-to-point xGMI links than fetching two remote rows per triplet.
"""
from __future__ import annotations

import numpy as np


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


class ReplicatedTableSync:
    """Delta all-reduce of a replicated table held in a torch tensor (cpu or cuda).

    ``sync()`` after each step makes every replica equal to start + sum of all ranks' deltas and
    re-arms the snapshot.  With world size 1 it is a no-op apart from refreshing the snapshot."""

    def __init__(self, table, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.table = table
        self.group = group
        self.start = table.clone()

    def sync(self):
        delta = self.table - self.start
        if self._dist.is_initialized() and self._dist.get_world_size(self.group) > 1:
            self._dist.all_reduce(delta, group=self.group)
        self.start.add_(delta)
        self.table.copy_(self.start)


# ---- graph models (LightGCN / NGCF / SimGCL): batch-sharded data parallelism -------------------------------------
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

    def __init__(self, group=None, device_index: int | None = None):
        import torch
        import torch.distributed as dist
        from . import capi
        self._torch, self._dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device("cuda", capi.current_device() if device_index is None else device_index)
        self._host_via_device = dist.get_backend(group) == "nccl"

    @classmethod
    def from_env(cls):
        """the process group ``init_from_env`` made, or None on one GPU (torch is not imported then)"""
        import sys
        dist = sys.modules.get("torch.distributed")
        if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        return cls()

    def share(self, n_rows: int) -> tuple[int, int]:
        """(offset, count) of this rank's contiguous share of a step's rows (counts differ by <= 1)"""
        lo, hi = user_block(n_rows, self.world, self.rank)
        return lo, hi - lo

    def all_reduce(self, buf):
        """sum a device buffer over the ranks in place; ordered after the kernels enqueued so far (default stream)"""
        self._dist.all_reduce(self._torch.as_tensor(buf, device=self.device), group=self.group)

    def all_reduce_host(self, arr: np.ndarray) -> np.ndarray:
        t = self._torch.from_numpy(np.ascontiguousarray(arr))
        if self._host_via_device:
            t = t.to(self.device)
        self._dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()


def init_from_env():
    """``python -m torch.distributed.run --nproc-per-node G -m qrec_amd.main <conf>``: one process per GPU.  Joins
    the process group (RCCL; gloo with every rank on device 0 under QREC_DIST_TEST_ONE_DEVICE=1, the functional test
    on 1-GPU boxes), binds the rank's device and gives every rank the same ``random`` / ``numpy.random`` streams --
    the reference never seeds, so rank 0's choice (QREC_SEED or the clock) is broadcast.  Returns the
    world size; 1 (and no torch import) when not launched that way."""
    import os
    import random
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1
    import time
    import torch
    import torch.distributed as dist
    from . import capi
    one_device = os.environ.get("QREC_DIST_TEST_ONE_DEVICE") == "1"
    local = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["QREC_DEVICE"] = str(local)
    torch.cuda.set_device(local)
    if one_device:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    capi.init(local)
    seed = torch.tensor([int(os.environ.get("QREC_SEED", time.time_ns() % (2 ** 31)))], dtype=torch.int64)
    if not one_device:
        seed = seed.cuda()
    dist.broadcast(seed, src=0)
    random.seed(int(seed.item())); np.random.seed(int(seed.item()) % (2 ** 32))
    return world


def is_output_rank() -> bool:
    """result / measure files are written by rank 0 only"""
    import sys
    dist = sys.modules.get("torch.distributed")
    return dist is None or not dist.is_initialized() or dist.get_rank() == 0
