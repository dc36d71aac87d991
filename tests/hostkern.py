"""TEST INFRASTRUCTURE: a host (numpy) emulation of the few C-ABI entry points that qrec_amd/dist.py's orchestration
calls -- same names and argument meaning as qrec_amd/capi.py, "device" buffers are numpy arrays addressed by their host
pointers.  It lets the world-size-2 gloo tests run the product's exchange protocol (row-id requests, row lookups,
return of updates, delta all-reduce) on CPU; the kernels themselves are tested on the GPU against this same statement.
Never imported by the product."""
import ctypes

import numpy as np

F32, F64, I32 = 0, 1, 2
_TILE = 1024


class DeviceBuffer:
    def __init__(self, shape, dtype):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.a = np.zeros(self.shape, self.dtype)
        self.nbytes = self.a.nbytes
        self.ptr = self.a.ctypes.data

    @classmethod
    def from_numpy(cls, a):
        b = cls(a.shape, a.dtype); b.a[...] = a
        return b

    def numpy(self, stream=None):
        return self.a.copy()

    def head(self, n, stream=None):
        return self.a.ravel()[:n].copy()


def device_ptr(x):
    return x.ptr if isinstance(x, DeviceBuffer) else int(x)


def _view(x, count, dtype):
    dtype = np.dtype(dtype)
    if count == 0:
        return np.empty(0, dtype)
    buf = (ctypes.c_char * (count * dtype.itemsize)).from_address(device_ptr(x))
    return np.frombuffer(buf, dtype=dtype, count=count)


class PinnedBuffer:
    def __init__(self, n, dtype):
        self.a = np.zeros(max(int(n), 1), dtype); self.nbytes = self.a.nbytes


def memcpy_d2h_async(pinned, src, nbytes, stream=None):
    pinned.a.view(np.uint8)[:nbytes] = _view(src, nbytes, np.uint8)


def memcpy_d2h(host, src, nbytes, stream=None):
    host.view(np.uint8).ravel()[:nbytes] = _view(src, nbytes, np.uint8)


def memcpy_h2d(dst, host, nbytes, stream=None):
    _view(dst, nbytes, np.uint8)[:] = np.ascontiguousarray(host).view(np.uint8).ravel()[:nbytes]


def memcpy_d2d(dst, src, nbytes, stream=None):
    _view(dst, nbytes, np.uint8)[:] = _view(src, nbytes, np.uint8)


def table_delta(table, start, delta, n, stream=None):
    _view(delta, n, np.float32)[:] = _view(table, n, np.float32) - _view(start, n, np.float32)


def table_apply(table, start, delta, n, stream=None):
    s = _view(start, n, np.float32)
    s += _view(delta, n, np.float32)
    _view(table, n, np.float32)[:] = s


def shard_rows(n_items, world, rank):
    return n_items // world + (1 if rank < n_items % world else 0)


def shard_plan_scratch_bytes(n_items, world):
    return 16


def shard_plan_batch(d_i, d_j, n, n_items, world, d_scratch, d_req_rows, d_counts, d_ci, d_cj, stream=None):
    """include/qrec_hip.h qrec_shard_plan_batch: distinct items grouped by owner (rank order), ascending local row"""
    i, j = _view(d_i, n, np.int32), _view(d_j, n, np.int32)
    items = np.unique(np.concatenate([i, j]))
    owner, row = items % world, items // world
    order = np.lexsort((row, owner))
    items, owner, row = items[order], owner[order], row[order]
    _view(d_counts, world, np.int32)[:] = np.bincount(owner, minlength=world)
    if n:
        _view(d_req_rows, items.size, np.int32)[:] = row
        slot = np.full(n_items, -1, np.int64); slot[items] = np.arange(items.size)
        _view(d_ci, n, np.int32)[:] = slot[i]
        _view(d_cj, n, np.int32)[:] = slot[j]


def shard_plan_epoch_scratch_bytes(n_items, world, n_batches):
    return 16


def shard_plan_epoch(d_i, d_j, d_bounds, n_batches, n, n_items, world, d_scratch, d_req_rows, d_req_off, d_counts, d_ci, d_cj, stream=None):
    """include/qrec_hip.h qrec_shard_plan_epoch: qrec_shard_plan_batch for every batch [bounds[b], bounds[b + 1])"""
    bounds, roff = _view(d_bounds, n_batches + 1, np.int64), _view(d_req_off, n_batches, np.int64)
    for b in range(n_batches):
        t0, nb = int(bounds[b]), int(bounds[b + 1] - bounds[b])
        shard_plan_batch(device_ptr(d_i) + 4 * t0, device_ptr(d_j) + 4 * t0, nb, n_items, world, d_scratch, device_ptr(d_req_rows) + 4 * int(roff[b]),
                         device_ptr(d_counts) + 4 * b * world, device_ptr(d_ci) + 4 * t0, device_ptr(d_cj) + 4 * t0)


class Event:
    """host emulation: every "enqueue" has already happened when it returns, so events order nothing"""
    def record(self, stream=None):
        pass

    def sync(self):
        pass


def stream_wait_event(stream, ev):
    pass


def gather_rows(table, ld, d_rows, n, d_out, stream=None):
    rows = _view(d_rows, n, np.int32)
    if n:
        t = _view(table, (int(rows.max()) + 1) * ld, np.float32).reshape(-1, ld)
        _view(d_out, n * ld, np.float32).reshape(n, ld)[:] = t[rows]


def scatter_add_row_deltas(table, ld, d_rows, n, d_fresh, d_sent, stream=None):
    rows = _view(d_rows, n, np.int32)
    if n:
        t = _view(table, (int(rows.max()) + 1) * ld, np.float32).reshape(-1, ld)
        dlt = _view(d_fresh, n * ld, np.float32).reshape(n, ld) - _view(d_sent, n * ld, np.float32).reshape(n, ld)
        np.add.at(t, rows, dlt)
