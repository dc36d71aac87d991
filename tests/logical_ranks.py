"""G logical ranks in ONE process on one device (SURVEY s8e "testing without a multi-GPU box"): every rank is a thread, a
collective is a rendezvous plus device-to-device copies.  ``ThreadComm`` has ``capi.Comm``'s interface; used by the GPU tests of the
multi-GPU layouts (tests/test_gpu_dist.py) and by the paired Recall@20 harness (tools/paired_recall.py).  Test infrastructure."""
import threading

import numpy as np

from qrec_amd import capi


class _Group:
    def __init__(self, world):
        self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world


class ThreadComm:
    """in-process fake collective (SURVEY s8e 'testing without a multi-GPU box'): every logical rank is a thread, a
    collective is a rendezvous + device-to-device copies.  capi.Comm's interface."""

    def __init__(self, group, rank):
        self.g, self.world, self.rank = group, group.world, rank

    def _swap(self, payload):
        capi.device_sync()
        self.g.slots[self.rank] = payload
        self.g.barrier.wait()
        everyone = list(self.g.slots)
        self.g.barrier.wait()
        return everyone

    def _done(self):
        capi.device_sync()
        self.g.barrier.wait()

    def alltoall_rows(self, send, send_rows, recv, recv_rows, row_bytes, stream=None):
        everyone = self._swap((capi.device_ptr(send) if send is not None else 0, [int(x) for x in send_rows]))
        off = 0
        for p, (ptr, rows) in enumerate(everyone):
            assert rows[self.rank] == int(recv_rows[p])
            nb = rows[self.rank] * row_bytes
            if nb:
                capi.memcpy_d2d(capi.device_ptr(recv) + off, ptr + sum(rows[:self.rank]) * row_bytes, nb)
            off += nb
        self._done()

    def sendrecv_segments(self, send, sends, recv, recvs, stream=None):
        everyone = self._swap((capi.device_ptr(send) if send is not None else 0, [tuple(int(x) for x in s) for s in sends]))
        for p, (ptr, their) in enumerate(everyone):
            to_me = [(o, nb) for q, o, nb in their if q == self.rank]            # what rank p sends me, in its list order
            mine = [(o, nb) for q, o, nb in recvs if q == p]                     # what I receive from rank p, in my list order
            assert [nb for _, nb in to_me] == [nb for _, nb in mine]
            for (so, nb), (ro, _) in zip(to_me, mine):
                if nb:
                    capi.memcpy_d2d(capi.device_ptr(recv) + ro, ptr + so, nb)
        self._done()

    def allgather(self, send, recv, count, dtype=capi.F32, stream=None):
        size = {capi.F32: 4, capi.F64: 8, capi.I32: 4}[dtype] * count
        for p, ptr in enumerate(self._swap(capi.device_ptr(send))):
            capi.memcpy_d2d(capi.device_ptr(recv) + p * size, ptr, size)
        self._done()

    def allreduce(self, buf, count, dtype=capi.F32, stream=None):
        npdt = {capi.F32: np.float32, capi.F64: np.float64, capi.I32: np.int32}[dtype]
        capi.device_sync()          # the copy below runs on the null stream: the rank's own (non-blocking) stream must have produced buf
        mine = np.empty(count, npdt); capi.memcpy_d2h(mine, buf, mine.nbytes)
        total = sum(self._swap(mine))
        capi.memcpy_h2d(buf, np.ascontiguousarray(total, dtype=npdt), mine.nbytes)
        self._done()

    def allreduce_pair(self, a, count_a, dtype_a, b, count_b, dtype_b, stream=None):
        self.allreduce(a, count_a, dtype_a, stream); self.allreduce(b, count_b, dtype_b, stream)


def run_ranks(world, fn, timeout=1800):
    """run ``fn(rank, group)`` on ``world`` threads; a failing rank releases the others from the barrier and its exception is raised here"""
    group, errors, out = _Group(world), [], [None] * world

    def body(rank):
        try:
            capi.init(0)
            out[rank] = fn(rank, group)
        except BaseException as e:      # noqa: BLE001
            errors.append(e); group.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=timeout)
    if errors:
        raise errors[0]
    return out
