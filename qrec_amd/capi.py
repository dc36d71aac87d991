"""ctypes binding of libqrec_hip.so (include/qrec_hip.h) -- the only door to the device.

There is deliberately NO CPU fallback: if the shared library is missing or a call fails,
this module raises.  numpy arrays cross the boundary as borrowed host pointers; device
memory is an opaque ``DeviceBuffer`` (or any raw device pointer, e.g. ``tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqrec_hip.so")

F32, F64, I32 = 0, 1, 2
HW_DEFAULT, HW_PLAIN_RMW, HW_SC1_RMW, HW_ATOMIC, HW_SC1_ATOMIC, HW_P_RMW, HW_PQ_RMW = 0, 1, 2, 3, 4, 5, 6

_vp, _i32, _i64, _u64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double

# name -> argtypes; every function returns int (0 = ok) unless listed in _RESTYPES
_SIGNATURES = {
    "qrec_version": [],
    "qrec_last_error": [],
    "qrec_device_count": [_vp],
    "qrec_init": [C.c_int],
    "qrec_device_info": [_vp, C.c_int, _vp, _vp, _vp, C.c_int],
    "qrec_malloc": [_i64, _vp],
    "qrec_free": [_vp],
    "qrec_memcpy_h2d": [_vp, _vp, _i64, _vp],
    "qrec_memcpy_d2h": [_vp, _vp, _i64, _vp],
    "qrec_memcpy_d2d": [_vp, _vp, _i64, _vp],
    "qrec_host_alloc": [_i64, _vp],
    "qrec_host_free": [_vp],
    "qrec_memcpy_d2h_async": [_vp, _vp, _i64, _vp],
    "qrec_memset": [_vp, C.c_int, _i64, _vp],
    "qrec_stream_create": [_vp],
    "qrec_stream_destroy": [_vp],
    "qrec_stream_sync": [_vp],
    "qrec_device_sync": [],
    "qrec_event_create": [_vp],
    "qrec_event_destroy": [_vp],
    "qrec_event_record": [_vp, _vp],
    "qrec_event_sync": [_vp],
    "qrec_stream_wait_event": [_vp, _vp],
    "qrec_event_elapsed_ms": [_vp, _vp, _vp],
    "qrec_mt_bpr_sample_epoch": [_vp, _vp, _vp, _i32, _i32, _vp],
    "qrec_mt_tbpr_sample_epoch": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "qrec_tbpr_sgd_ordered": [_vp, _vp, C.c_int, _i32, _i32, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _vp, _vp, _vp],
    "qrec_mt_sbpr_sample_epoch": [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "qrec_sbpr_sgd_ordered": [_vp, _vp, _vp, C.c_int, _i32, _i32, _vp, _i64, _f64, _f64, _f64, _f64, _vp, _vp, _vp],
    "qrec_mt_shuffle": [_vp, _i64, _vp],
    "qrec_mt_sample_range": [_vp, _i64, _i64, _vp],
    "qrec_mt_pairwise_sample_epoch": [_vp, _vp, _i64, _vp, _vp, _i32, _vp],
    "qrec_gather_pairs": [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "qrec_philox_bpr_sample": [_vp, _vp, _vp, _i64, _i32, _u64, _u64, _vp, _vp],
    "qrec_bpr_sgd_ordered": [_vp, _vp, C.c_int, _i32, _i32, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _vp, _vp],
    "qrec_bpr_exact_width": [C.c_int, _i32, _vp],
    "qrec_bpr_exact_schedule": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "qrec_bpr_sgd_scheduled": [_vp, _vp, C.c_int, _i32, _i32, _vp, _vp, _i64, _i32, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp],
    "qrec_bpr_exact_kind": [C.c_int, _i32, _i32, _vp, _vp],
    "qrec_bpr_exact_schedule_reg": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "qrec_bpr_exact_expand": [_vp, _vp, _i64, _i32, _vp, _vp],
    "qrec_bpr_sgd_scheduled_wide": [_vp, _vp, _i64, _i64, C.c_int, _i32, _i32, _vp, _i64, _i32, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp],
    "qrec_bpr_sgd_hogwild": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _vp, C.c_int, _vp, _vp],
    "qrec_bpr_sgd_hogwild_item_major": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp, C.c_int, _vp, _vp],
    "qrec_epoch_close": [_vp, _i64, _vp, _i64, C.c_int, _i32, _vp, _vp, _f64, _f64, _f64, _f64, _vp, _i64, _vp],
    "qrec_epoch_sums": [_vp, _i64, _vp, _i64, C.c_int, _i32, _vp, _vp, _vp],
    "qrec_epoch_decide": [_vp, _vp, _f64, _f64, _f64, _f64, _vp, _i64, _vp],
    "qrec_epoch_sum_table": [_vp, _i64, C.c_int, _i32, _vp, C.c_int, _vp, _vp],
    "qrec_mf_sgd_ordered": [_vp, _vp, C.c_int, _i32, _i32, _vp, _vp, _vp, _i64, _f64, _vp, C.c_int, _f64, _f64, _vp, _vp, _f64, _f64, _vp],
    "qrec_svdpp_sgd_ordered": [_vp, _vp, _vp, _vp, _vp, C.c_int, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64,
                               _f64, _f64, _f64, _vp, _vp],
    "qrec_sumsq": [_vp, C.c_int, _i64, _i32, _i32, _vp, _vp],
    "qrec_spmm_csr": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    "qrec_mark_batch_rows": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "qrec_bpr_batch_loss_grad": [_vp, _f32, _i32, _i64, _i32, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _i64, _vp],
    "qrec_subgraph_values": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "qrec_ordered_scatter_workspace_bytes": [_i64, _i32, _vp],
    "qrec_scatter_add_rows_ordered": [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp],
    "qrec_adam_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
    "qrec_perturb_rows": [_vp, _vp, _i64, _i32, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _vp, _i32, _i64, _vp],
    "qrec_perturb_two_views": [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp],
    "qrec_info_nce_workspace_bytes": [_i32, _i32, _vp],
    "qrec_gate_fwd": [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "qrec_gate_bwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _i32, _vp],
    "qrec_channel_attention_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp],
    "qrec_channel_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp],
    "qrec_channel_attention_scratch_bytes": [_vp],
    "qrec_hss_scratch_bytes": [_i64, _vp],
    "qrec_hss_loss_grad": [_vp, _vp, _i64, _i32, _i32] + [_vp] * 10 + [_f32, _vp, _vp, _vp, _vp, _vp],
    "qrec_random_permutations_scratch_bytes": [_i64, _i32, _vp],
    "qrec_random_permutations": [_i64, _i32, _u64, _u64, _vp, _vp, _vp, _vp],
    "qrec_small_permutations": [_i32, _i32, _u64, _u64, _vp, _vp, _vp],
    "qrec_l2norm_rows_accum": [_vp, _i64, _i32, _vp, _vp, _vp],
    "qrec_l2norm_rows_bwd": [_vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "qrec_scale_copy": [_vp, _vp, _i64, _f32, _vp],
    "qrec_sept_ssl_workspace_bytes": [_i32, _i32, _i32, _vp],
    "qrec_sept_ssl_loss_grad": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "qrec_info_nce_loss_grad": [_vp, _vp, _f32, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "qrec_ngcf_dense_fwd": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp],
    "qrec_compact_marked_rows": [_vp, _i64, _vp, _vp, _i32, _vp],
    "qrec_unique_per_batch": [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "qrec_mark_compact_batch_rows": [_vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _vp, _i32, _vp],
    "qrec_ngcf_activate": [_vp, _i64, _i32, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i64, _vp],
    "qrec_ngcf_layer_bwd": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "qrec_ngcf_wgrad_partial_bytes": [_i64, _i32, _vp],
    "qrec_copy_cols": [_vp, _i32, _vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp],
    "qrec_zero_rows": [_vp, _i32, _vp, _vp, _i32, _vp],
    "qrec_score_topk_scratch_bytes": [C.c_int, _i32, _i32, _i32, _i32, _vp],
    "qrec_score_topk": [_vp, _vp, C.c_int, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "qrec_rank_hits": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "qrec_buir_batch_loss_grad": [_vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "qrec_buir_wgrad_scratch_bytes": [_i32, _vp],
    "qrec_buir_wgrad": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "qrec_ema_update": [_vp, _vp, _f32, _i64, _vp],
    "qrec_mt_data_split": [_vp, _i64, _f64, _vp],
    "qrec_ratings_load": [C.c_char_p, C.c_char_p, _i32, _i32, _i32, _i32, _i32, _f64, _vp],
    "qrec_ratings_rows": [_vp],
    "qrec_ratings_count": [_vp, _i32],
    "qrec_ratings_names_bytes": [_vp, _i32],
    "qrec_ratings_copy": [_vp, _vp, _vp, _vp],
    "qrec_ratings_names": [_vp, _i32, _vp],
    "qrec_ratings_free": [_vp],
    "qrec_comm_library": [_vp, C.c_int, _vp],
    "qrec_comm_unique_id": [_vp],
    "qrec_comm_init": [_i32, _i32, _vp, _vp],
    "qrec_comm_destroy": [_vp],
    "qrec_comm_info": [_vp, _vp, _vp],
    "qrec_comm_query": [_vp, _vp, _vp, _vp],
    "qrec_allreduce": [_vp, _vp, _i64, C.c_int, _vp],
    "qrec_allreduce_pair": [_vp, _vp, _i64, C.c_int, _vp, _i64, C.c_int, _vp],
    "qrec_allgather": [_vp, _vp, _vp, _i64, C.c_int, _vp],
    "qrec_reduce_scatter": [_vp, _vp, _vp, _i64, C.c_int, _vp],
    "qrec_alltoall_rows": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "qrec_sendrecv_segments": [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "qrec_dist_epoch_pre": [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "qrec_dist_epoch_post": [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _f64, _f64, _f64, _f64, _vp, _i64, _vp],
    "qrec_table_delta": [_vp, _vp, _vp, _i64, _vp],
    "qrec_table_apply": [_vp, _vp, _vp, _i64, _vp],
    "qrec_shard_rows": [_i64, _i32, _i32, _vp],
    "qrec_shard_plan_scratch_bytes": [_i64, _i32, _vp],
    "qrec_shard_plan_batch": [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "qrec_shard_plan_epoch_scratch_bytes": [_i64, _i32, _i32, _vp],
    "qrec_shard_plan_epoch": [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "qrec_gather_rows": [_vp, _i32, _vp, _i64, _vp, _vp],
    "qrec_rows_gather_owned": [_vp, _i32, _i64, _i64, _vp, _i64, _vp, _vp],
    "qrec_rows_scatter_add_owned": [_vp, _i32, _i64, _i64, _vp, _i64, _vp, _vp],
    "qrec_batch_rows_gather": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _i64, _vp, _vp],
    "qrec_batch_rows_scatter_add": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _i64, _vp, _vp],
    "qrec_scatter_add_row_deltas": [_vp, _i32, _vp, _i64, _vp, _vp, _vp],
}
_RESTYPES = {"qrec_last_error": C.c_char_p, "qrec_ratings_rows": C.c_int64, "qrec_ratings_count": C.c_int32,
             "qrec_ratings_names_bytes": C.c_int64, "qrec_ratings_free": None}
ERR_UNSUPPORTED = -4

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class QRecError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libqrec_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """dlopen libqrec_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C qrec_amd/csrc).  There is no CPU fallback for the hot path.")
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def _check(code: int):
    if code != 0:
        raise QRecError(code, load().qrec_last_error().decode(errors="replace"))


def _hp(a: np.ndarray):
    """borrowed host pointer of a C-contiguous numpy array"""
    if not (isinstance(a, np.ndarray) and a.flags.c_contiguous):
        raise TypeError("need a C-contiguous numpy array")
    return a.ctypes.data_as(C.c_void_p)


def _req(a, dtype, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.dtype(dtype) and a.flags.c_contiguous):
        raise TypeError(f"{name}: need C-contiguous {np.dtype(dtype)}, got "
                        f"{getattr(a, 'dtype', type(a))}")
    return a


# ---- runtime ------------------------------------------------------------------------------
_initialised = False
_device = 0


def init(device: int | None = None):
    """Select the device (env QREC_DEVICE, default 0).  Never called at import or from a
    model constructor: QRec forks per CV fold after constructing models (QRec.py:76-89)."""
    global _initialised, _device
    if device is None:
        device = int(os.environ.get("QREC_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    # a launcher that pins one GPU per process (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES) leaves every rank with a
    # single visible device, index 0, while LOCAL_RANK still counts up: take the rank's device modulo what is visible
    n = C.c_int(0)
    _check(load().qrec_device_count(C.byref(n)))
    if n.value > 0 and device >= n.value:
        device %= n.value
    _check(load().qrec_init(device))
    _initialised, _device = True, device


def current_device() -> int:
    ensure_init()
    return _device


def ensure_init():
    if not _initialised:
        init()


def device_count() -> int:
    n = C.c_int(0)
    _check(load().qrec_device_count(C.byref(n)))
    return n.value


def device_info() -> dict:
    name = C.create_string_buffer(256); arch = C.create_string_buffer(64)
    ncu = C.c_int(0); hbm = C.c_int64(0)
    _check(load().qrec_device_info(name, 256, C.byref(ncu), C.byref(hbm), arch, 64))
    return dict(name=name.value.decode(), arch=arch.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value)


def device_ptr(x) -> int:
    """device address of a DeviceBuffer / torch tensor / raw int -- for pointer arithmetic into a larger buffer"""
    return int(_dp(x))


def memcpy_d2d(dst, src, nbytes: int, stream=None):
    _check(load().qrec_memcpy_d2d(_dp(dst), _dp(src), nbytes, _sh(stream)))


def memset(dst, byte: int, nbytes: int, stream=None):
    _check(load().qrec_memset(_dp(dst), byte, nbytes, _sh(stream)))


def device_sync():
    _check(load().qrec_device_sync())


def stream_wait_event(stream, ev: "Event"):
    """make `stream` (None = the null stream) wait on the device for `ev`"""
    _check(load().qrec_stream_wait_event(_sh(stream), ev.handle))


class Stream:
    def __init__(self):
        ensure_init()
        h = C.c_void_p()
        _check(load().qrec_stream_create(C.byref(h)))
        self.handle = h.value

    def sync(self):
        _check(load().qrec_stream_sync(self.handle))

    def wait_event(self, ev: "Event"):
        _check(load().qrec_stream_wait_event(self.handle, ev.handle))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().qrec_stream_destroy(self.handle); self.handle = None
        except Exception:
            pass


def _sh(stream):
    if stream is None:
        return None
    return stream.handle if isinstance(stream, Stream) else int(stream)


class Event:
    def __init__(self):
        ensure_init()
        h = C.c_void_p()
        _check(load().qrec_event_create(C.byref(h)))
        self.handle = h.value

    def record(self, stream=None):
        _check(load().qrec_event_record(self.handle, _sh(stream)))

    def sync(self):
        _check(load().qrec_event_sync(self.handle))

    def elapsed_ms_since(self, start: "Event") -> float:
        ms = C.c_float(0)
        _check(load().qrec_event_elapsed_ms(start.handle, self.handle, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().qrec_event_destroy(self.handle); self.handle = None
        except Exception:
            pass


class DeviceBuffer:
    """Owning handle on device memory with a numpy-like shape/dtype tag."""

    def __init__(self, shape, dtype):
        ensure_init()
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        _check(load().qrec_malloc(self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    @classmethod
    def from_numpy(cls, a: np.ndarray, stream=None) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.shape, a.dtype)
        b.upload(a, stream)
        return b

    @classmethod
    def zeros(cls, shape, dtype, stream=None) -> "DeviceBuffer":
        b = cls(shape, dtype)
        b.fill_bytes(0, stream)
        return b

    @property
    def __cuda_array_interface__(self):
        """zero-copy view for torch.as_tensor(buf, device="cuda") (multi-GPU collectives)"""
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False),
                "version": 2, "strides": None}

    def head_view(self, n: int):
        """borrowed view of the first ``n`` elements, for torch.as_tensor (a collective on part of a buffer)"""
        owner, itf = self, {"shape": (n,), "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2, "strides": None}
        return type("DeviceView", (), {"__cuda_array_interface__": itf, "owner": owner})()

    def upload(self, a: np.ndarray, stream=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError(f"upload size mismatch: {a.nbytes} vs {self.nbytes}")
        _check(load().qrec_memcpy_h2d(self.ptr, _hp(a), self.nbytes, _sh(stream)))

    def numpy(self, stream=None) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        _check(load().qrec_memcpy_d2h(_hp(out), self.ptr, self.nbytes, _sh(stream)))
        return out

    def head(self, n: int, stream=None) -> np.ndarray:
        """the first ``n`` elements (flat) -- small read-backs of a larger buffer"""
        out = np.empty(n, dtype=self.dtype)
        _check(load().qrec_memcpy_d2h(_hp(out), self.ptr, out.nbytes, _sh(stream)))
        return out

    def read_rows(self, row0: int, n_rows: int, stream=None) -> np.ndarray:
        """rows [row0, row0 + n_rows) of a 2-D buffer, without reading the rest back"""
        width = self.shape[1]
        out = np.empty((n_rows, width), dtype=self.dtype)
        _check(load().qrec_memcpy_d2h(_hp(out), self.ptr + row0 * width * self.dtype.itemsize, out.nbytes, _sh(stream)))
        return out

    def write_rows(self, row0: int, a: np.ndarray, stream=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        _check(load().qrec_memcpy_h2d(self.ptr + row0 * self.shape[1] * self.dtype.itemsize, _hp(a), a.nbytes, _sh(stream)))

    def upload_head(self, a: np.ndarray, stream=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes > self.nbytes:
            raise ValueError("upload_head: larger than the buffer")
        _check(load().qrec_memcpy_h2d(self.ptr, _hp(a), a.nbytes, _sh(stream)))

    def copy_from(self, other: "DeviceBuffer", stream=None, nbytes: int | None = None):
        """device-to-device copy of the whole buffer, or of the first ``nbytes`` bytes of both"""
        if nbytes is None:
            if other.nbytes != self.nbytes:
                raise ValueError("copy_from size mismatch")
            nbytes = self.nbytes
        elif nbytes > min(self.nbytes, other.nbytes):
            raise ValueError("copy_from: more bytes than the buffers hold")
        if nbytes:
            _check(load().qrec_memcpy_d2d(self.ptr, other.ptr, nbytes, _sh(stream)))

    def fill_bytes(self, byte: int = 0, stream=None):
        _check(load().qrec_memset(self.ptr, byte, self.nbytes, _sh(stream)))

    def free(self):
        if getattr(self, "ptr", 0):
            load().qrec_free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceSlice(DeviceBuffer):
    """Non-owning window [offset, offset + prod(shape)) (elements) of a DeviceBuffer: several small tables carved out
    of one allocation so that one kernel launch can cover all of them (e.g. the four NGCF weight matrices in Adam)."""

    def __init__(self, owner: DeviceBuffer, offset_elems: int, shape):
        self.owner = owner
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = owner.dtype
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if offset_elems < 0 or offset_elems * self.dtype.itemsize + self.nbytes > owner.nbytes:
            raise ValueError("DeviceSlice outside its owner")
        self.ptr = owner.ptr + offset_elems * self.dtype.itemsize

    def free(self):           # the owner frees
        self.ptr = 0


def _dp(x):
    """device pointer of a DeviceBuffer / int / object with data_ptr() (torch tensor)"""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


# ---- exact sampler (host) ---------------------------------------------------------------------
def state_from_python(state) -> np.ndarray:
    """random.getstate() -> uint32[625]"""
    return np.array(state[1], dtype=np.uint32)


def state_to_python(words: np.ndarray, gauss_next=None):
    return (3, tuple(int(x) for x in words), gauss_next)


def mt_bpr_sample_epoch(state625: np.ndarray, pos_indptr, pos_indices, n_items: int) -> np.ndarray:
    _req(state625, np.uint32, "state625"); _req(pos_indptr, np.int64, "pos_indptr")
    _req(pos_indices, np.int32, "pos_indices")
    j = np.empty(pos_indices.size, dtype=np.int32)
    _check(load().qrec_mt_bpr_sample_epoch(_hp(state625), _hp(pos_indptr), _hp(pos_indices),
                                           pos_indptr.size - 1, n_items, _hp(j)))
    return j


def mt_tbpr_sample_epoch(state625: np.ndarray, pos_indptr, pos_items, n_items: int, joint, weak, strong):
    """TBPR.py:131-158 on the CPython stream: the epoch's chained (u, a, b) updates.  joint / weak / strong:
    (indptr int64, items int32) CSR over users of the three candidate lists."""
    _req(state625, np.uint32, "state625"); _req(pos_indptr, np.int64, "pos_indptr"); _req(pos_items, np.int32, "pos_items")
    for name, (ptr, items) in (("joint", joint), ("weak", weak), ("strong", strong)):
        _req(ptr, np.int64, name + " indptr"); _req(items, np.int32, name + " items")
        if ptr.size != pos_indptr.size:
            raise ValueError(f"{name} list: one row per user expected")
    cap = 4 * int(pos_items.size)
    u, a, b = (np.empty(max(cap, 1), dtype=np.int32) for _ in range(3))
    n = C.c_int64(0)
    _check(load().qrec_mt_tbpr_sample_epoch(_hp(state625), _hp(pos_indptr), _hp(pos_items), pos_indptr.size - 1, n_items,
                                            _hp(joint[0]), _hp(joint[1]), _hp(weak[0]), _hp(weak[1]), _hp(strong[0]),
                                            _hp(strong[1]), cap, _hp(u), _hp(a), _hp(b), C.byref(n)))
    return u[:n.value].copy(), a[:n.value].copy(), b[:n.value].copy()


def tbpr_sgd_ordered(d_P, d_Q, dtype: int, d: int, ld: int, d_u, d_a, d_b, n: int, lr: float, regU: float, regI: float,
                     d_sums_in, d_loss2, stream=None):
    """TBPR.py:40-48,157-159 over the chained triplets, strictly in order (see include/qrec_hip.h)"""
    _check(load().qrec_tbpr_sgd_ordered(_dp(d_P), _dp(d_Q), dtype, d, ld, _dp(d_u), _dp(d_a), _dp(d_b), n, lr, regU, regI,
                                        _dp(d_sums_in), _dp(d_loss2), _sh(stream)))


def mt_sbpr_sample_epoch(state625: np.ndarray, ps_users, pos_indptr, pos_items, n_items: int, fp_indptr, fp_items, fp_counts, item_key_user,
                         is_key) -> np.ndarray:
    """SBPR.py:37-55,69-72 on the CPython stream: rows (u, i, k or -1, j, Suk) of one epoch, int32 [n, 5] (include/qrec_hip.h).
    ``is_key`` (uint8 per user) is updated in place."""
    _req(state625, np.uint32, "state625"); _req(ps_users, np.int32, "ps_users"); _req(pos_indptr, np.int64, "pos_indptr")
    _req(pos_items, np.int32, "pos_items"); _req(fp_indptr, np.int64, "fp_indptr"); _req(fp_items, np.int32, "fp_items")
    _req(fp_counts, np.int32, "fp_counts"); _req(item_key_user, np.int32, "item_key_user"); _req(is_key, np.uint8, "is_key")
    n_users = pos_indptr.size - 1
    if fp_indptr.size != pos_indptr.size or is_key.size != n_users or item_key_user.size != n_items or fp_counts.size != fp_items.size:
        raise ValueError("mt_sbpr_sample_epoch: one FPSet row and one key flag per user, one name link per item expected")
    cap = int(pos_items.size)
    rows = np.empty((max(cap, 1), 5), dtype=np.int32)
    n = C.c_int64(0)
    _check(load().qrec_mt_sbpr_sample_epoch(_hp(state625), _hp(ps_users), ps_users.size, _hp(pos_indptr), _hp(pos_items), n_users, n_items,
                                            _hp(fp_indptr), _hp(fp_items), _hp(fp_counts), _hp(item_key_user), _hp(is_key), cap, _hp(rows), C.byref(n)))
    return rows[:n.value].copy()


def sbpr_sgd_ordered(d_P, d_Q, d_bias, dtype: int, d: int, ld: int, d_rows, n: int, lr: float, regU: float, regI: float, bias_sumsq: float,
                     d_sums_in, d_loss2, stream=None):
    """SBPR.py:41-74 over the epoch's rows, strictly in order (see include/qrec_hip.h)"""
    _check(load().qrec_sbpr_sgd_ordered(_dp(d_P), _dp(d_Q), _dp(d_bias), dtype, d, ld, _dp(d_rows), n, lr, regU, regI, bias_sumsq,
                                        _dp(d_sums_in), _dp(d_loss2), _sh(stream)))


def mt_data_split(state625: np.ndarray, n: int, ratio: float) -> np.ndarray:
    """util/dataSplit.py:14-22: mask[k] = (random() < ratio) for n consecutive draws of the CPython stream"""
    _req(state625, np.uint32, "state625")
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    _check(load().qrec_mt_data_split(_hp(state625), n, ratio, _hp(mask)))
    return mask[:n].astype(bool)


def ratings_load(path: str, delims: str | None, col_user: int, col_item: int, col_rating: int, skip_header: bool,
                 binarize: bool, threshold: float):
    """util/io.py:31-76 natively.  Returns (user_idx int32[n], item_idx int32[n], rating float64[n], user_names,
    item_names) with ids in first-appearance order, or None when the file needs CPython's own parsing rules."""
    lib = load()
    h = C.c_void_p()
    rc = lib.qrec_ratings_load(os.fsencode(path), delims.encode() if delims else None, col_user, col_item, col_rating,
                               int(skip_header), int(binarize), threshold, C.byref(h))
    if rc == ERR_UNSUPPORTED:
        return None
    _check(rc)
    try:
        n = lib.qrec_ratings_rows(h)
        u = np.empty(n, np.int32); i = np.empty(n, np.int32); r = np.empty(n, np.float64)
        _check(lib.qrec_ratings_copy(h, _hp(u), _hp(i), _hp(r)))
        names = []
        for which in (0, 1):
            buf = C.create_string_buffer(max(int(lib.qrec_ratings_names_bytes(h, which)), 1))
            _check(lib.qrec_ratings_names(h, which, buf))
            cnt = lib.qrec_ratings_count(h, which)
            names.append(buf.raw[:lib.qrec_ratings_names_bytes(h, which)].decode("ascii").split("\n") if cnt else [])
        return u, i, r, names[0], names[1]
    finally:
        lib.qrec_ratings_free(h)


def mt_shuffle(state625: np.ndarray, n: int, perm: np.ndarray | None = None):
    _req(state625, np.uint32, "state625")
    if perm is not None:
        _req(perm, np.int64, "perm")
        if perm.size != n:
            raise ValueError("perm size mismatch")
    _check(load().qrec_mt_shuffle(_hp(state625), n, _hp(perm) if perm is not None else None))
    return perm


def mt_pairwise_sample_epoch(state625, row_user, rated_indptr, rated_sorted, n_items: int) -> np.ndarray:
    _req(state625, np.uint32, "state625"); _req(row_user, np.int32, "row_user")
    _req(rated_indptr, np.int64, "rated_indptr"); _req(rated_sorted, np.int32, "rated_sorted")
    neg = np.empty(row_user.size, dtype=np.int32)
    _check(load().qrec_mt_pairwise_sample_epoch(_hp(state625), _hp(row_user), row_user.size,
                                                _hp(rated_indptr), _hp(rated_sorted), n_items, _hp(neg)))
    return neg


# ---- device ops ---------------------------------------------------------------------------------
def philox_bpr_sample(d_indptr, d_sorted, d_row_user, n: int, n_items: int, seed: int, epoch: int,
                      d_j_out, stream=None):
    _check(load().qrec_philox_bpr_sample(_dp(d_indptr), _dp(d_sorted), _dp(d_row_user), n, n_items,
                                         seed & (2**64 - 1), epoch & (2**64 - 1), _dp(d_j_out), _sh(stream)))


def gather_pairs(d_perm, d_u, d_i, n: int, d_u_out, d_i_out, stream=None):
    _check(load().qrec_gather_pairs(_dp(d_perm), _dp(d_u), _dp(d_i), n, _dp(d_u_out), _dp(d_i_out), _sh(stream)))


def bpr_sgd_ordered(d_P, d_Q, dtype: int, d: int, ld: int, d_u, d_i, d_j, n: int, lr: float,
                    regU: float, regI: float, d_loss, stream=None):
    _check(load().qrec_bpr_sgd_ordered(_dp(d_P), _dp(d_Q), dtype, d, ld, _dp(d_u), _dp(d_i), _dp(d_j),
                                       n, lr, regU, regI, _dp(d_loss), _sh(stream)))


EXACT_MAX_WIDTH, EXACT_SCRATCH_WORDS, EXACT_XLOG_PAD, EXACT_WIDE_PAD = 16, 130, 16 * 256 + 64, 16


def bpr_exact_width(dtype: int, d: int) -> int:
    w = _i32(0)
    _check(load().qrec_bpr_exact_width(dtype, d, C.byref(w)))
    return w.value


def bpr_exact_schedule(u: np.ndarray, i: np.ndarray, j: np.ndarray, n_users: int, n_items: int, width: int, registers: bool = False):
    """static schedule of an epoch's triplets (include/qrec_hip.h): (entries int32[n, 8] step-major, step_off int32[n_steps + 1]).
    ``registers``: the schedule of the four-triplets-per-wavefront kernel (qrec_bpr_exact_schedule_reg, kind 1)."""
    _req(u, np.int32, "u"); _req(i, np.int32, "i"); _req(j, np.int32, "j")
    n = int(u.size)
    entries = np.empty((max(n, 1), 8), dtype=np.int32)
    off = np.empty((3 * n if registers else n) + 2, dtype=np.int32)      # the register schedule may leave steps empty: <= 3 n steps
    steps = C.c_int64(0)
    f = load().qrec_bpr_exact_schedule_reg if registers else load().qrec_bpr_exact_schedule
    _check(f(_hp(u), _hp(i), _hp(j), n, n_users, n_items, width, _hp(entries), _hp(off), C.byref(steps)))
    return entries[:n], off[:steps.value + 1].copy()


def bpr_sgd_scheduled(d_P, d_Q, dtype: int, d: int, ld: int, d_entries, d_step_off, n_steps: int, width: int, n: int, lr: float,
                      regU: float, regI: float, d_xlog, d_scratch, d_loss, stream=None):
    _check(load().qrec_bpr_sgd_scheduled(_dp(d_P), _dp(d_Q), dtype, d, ld, _dp(d_entries), _dp(d_step_off), n_steps, width, n, lr, regU,
                                         regI, _dp(d_xlog), _dp(d_scratch), _dp(d_loss), _sh(stream)))


def bpr_exact_kind(dtype: int, ld: int, width: int) -> tuple[int, int]:
    """(kind, slots): which kernel executes an order-exact epoch of (dtype, ld, width) -- 0 one triplet per wavefront
    (bpr_exact_schedule + bpr_sgd_scheduled), 1 four per wavefront (bpr_exact_schedule(registers=True) + bpr_exact_expand +
    bpr_sgd_scheduled_wide) -- and the slots per step of the wide schedule layout (0 for kind 0)"""
    k, s = _i32(0), _i32(0)
    _check(load().qrec_bpr_exact_kind(dtype, ld, width, C.byref(k), C.byref(s)))
    return k.value, s.value


def bpr_exact_expand(d_entries, d_step_off, n_steps: int, slots: int, d_wide, stream=None):
    _check(load().qrec_bpr_exact_expand(_dp(d_entries), _dp(d_step_off), n_steps, slots, _dp(d_wide), _sh(stream)))


def bpr_sgd_scheduled_wide(d_P, d_Q, n_users: int, n_items: int, dtype: int, d: int, ld: int, d_wide, n_steps: int, slots: int,
                           n: int, lr: float, regU: float, regI: float, d_xlog, d_scratch, d_loss, stream=None):
    _check(load().qrec_bpr_sgd_scheduled_wide(_dp(d_P), _dp(d_Q), n_users, n_items, dtype, d, ld, _dp(d_wide), n_steps, slots, n, lr,
                                              regU, regI, _dp(d_xlog), _dp(d_scratch), _dp(d_loss), _sh(stream)))


def _table_rows(buf, ld: int) -> int:
    """rows of a [rows][ld] fp32 table held in a DeviceBuffer"""
    return int(buf.nbytes // (4 * ld))


def bpr_sgd_hogwild(d_P, d_Q, d: int, ld: int, d_u, d_i, d_j, n: int, chunk: int, grid_groups: int,
                    lr: float, regU: float, regI: float, d_loss, variant: int = HW_DEFAULT, stream=None,
                    d_driver_state=None, p_rows: int | None = None, q_rows: int | None = None):
    """``p_rows`` / ``q_rows``: rows of the tables when they are not whole DeviceBuffers (a shard's row cache)"""
    _check(load().qrec_bpr_sgd_hogwild(_dp(d_P), _dp(d_Q), p_rows or _table_rows(d_P, ld), q_rows or _table_rows(d_Q, ld), d, ld, _dp(d_u), _dp(d_i), _dp(d_j), n,
                                       chunk, grid_groups, lr, regU, regI, _dp(d_loss), variant,
                                       _dp(d_driver_state), _sh(stream)))


MF_BASIC, MF_PMF, MF_SVD, MF_EE = 0, 1, 2, 3


def mf_sgd_ordered(d_P, d_Q, dtype: int, d: int, ld: int, d_u, d_i, d_rating, n: int, lr: float,
                   d_loss, stream=None, variant: int = MF_BASIC, regU: float = 0.0, regI: float = 0.0, d_Bu=None,
                   d_Bi=None, regB: float = 0.0, global_mean: float = 0.0):
    _check(load().qrec_mf_sgd_ordered(_dp(d_P), _dp(d_Q), dtype, d, ld, _dp(d_u), _dp(d_i), _dp(d_rating),
                                      n, lr, _dp(d_loss), variant, regU, regI, _dp(d_Bu), _dp(d_Bi), regB,
                                      global_mean, _sh(stream)))


def svdpp_sgd_ordered(d_P, d_Q, d_Y, d_Bu, d_Bi, dtype: int, d: int, ld: int, d_rated_indptr, d_rated_items, d_u, d_i,
                      d_rating, n: int, lr: float, regU: float, regI: float, regB: float, regY: float, global_mean: float,
                      d_loss, stream=None):
    _check(load().qrec_svdpp_sgd_ordered(_dp(d_P), _dp(d_Q), _dp(d_Y), _dp(d_Bu), _dp(d_Bi), dtype, d, ld, _dp(d_rated_indptr),
                                         _dp(d_rated_items), _dp(d_u), _dp(d_i), _dp(d_rating), n, lr, regU, regI, regB, regY,
                                         global_mean, _dp(d_loss), _sh(stream)))


def sumsq(d_x, dtype: int, rows: int, d: int, ld: int, d_out, stream=None):
    _check(load().qrec_sumsq(_dp(d_x), dtype, rows, d, ld, _dp(d_out), _sh(stream)))


def score_topk_scratch_bytes(dtype: int, n_items: int, n_batch_users: int, ld: int, N: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_score_topk_scratch_bytes(dtype, n_items, n_batch_users, ld, N, C.byref(out)))
    return out.value


def score_topk(d_U, d_V, dtype: int, d: int, ld: int, n_items: int, d_user_ids, n_batch_users: int,
               d_rated_indptr, d_rated_items, N: int, d_scratch, d_ids_out, d_scores_out, stream=None):
    _check(load().qrec_score_topk(_dp(d_U), _dp(d_V), dtype, d, ld, n_items, _dp(d_user_ids), n_batch_users,
                                  _dp(d_rated_indptr), _dp(d_rated_items), N, _dp(d_scratch), _dp(d_ids_out),
                                  _dp(d_scores_out), _sh(stream)))


def epoch_sums(d_P, p_rows: int, d_Q, q_rows: int, dtype: int, ld: int, d_stats, d_state=None, stream=None):
    _check(load().qrec_epoch_sums(_dp(d_P), p_rows, _dp(d_Q), q_rows, dtype, ld, _dp(d_stats), _dp(d_state), _sh(stream)))


def epoch_sum_table(d_X, rows: int, dtype: int, ld: int, d_stats, slot: int, d_state=None, stream=None):
    _check(load().qrec_epoch_sum_table(_dp(d_X), rows, dtype, ld, _dp(d_stats), slot, _dp(d_state), _sh(stream)))


class PinnedBuffer:
    """page-locked host array (qrec_host_alloc); ``a`` is a numpy view of it"""

    def __init__(self, n: int, dtype):
        ensure_init()
        self.dtype = np.dtype(dtype)
        self.nbytes = max(int(n), 1) * self.dtype.itemsize
        h = C.c_void_p()
        _check(load().qrec_host_alloc(self.nbytes, C.byref(h)))
        self.ptr = h.value
        self.a = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=self.dtype)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.a = None
                load().qrec_host_free(C.c_void_p(self.ptr)); self.ptr = None
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


def memcpy_d2h_async(pinned: PinnedBuffer, src, nbytes: int, stream=None):
    """device -> page-locked host buffer, enqueue only: read ``pinned.a`` after an event recorded behind this call"""
    if nbytes > pinned.nbytes:
        raise ValueError("memcpy_d2h_async: pinned buffer too small")
    _check(load().qrec_memcpy_d2h_async(C.c_void_p(pinned.ptr), _dp(src), nbytes, _sh(stream)))


def memcpy_d2h(host: np.ndarray, src, nbytes: int, stream=None):
    """device -> caller-owned host array (synchronous on ``stream``)"""
    _check(load().qrec_memcpy_d2h(_hp(host), _dp(src), nbytes, _sh(stream)))


def memcpy_h2d(dst, host: np.ndarray, nbytes: int, stream=None):
    _check(load().qrec_memcpy_h2d(_dp(dst), _hp(host), nbytes, _sh(stream)))


def epoch_decide(d_stats, d_state, regU: float, regI: float, max_lr: float, tol: float, d_log=None,
                 log_capacity: int = 0, stream=None):
    _check(load().qrec_epoch_decide(_dp(d_stats), _dp(d_state), regU, regI, max_lr, tol, _dp(d_log), log_capacity,
                                    _sh(stream)))


def buir_batch_loss_grad(d_S_on, d_S_tar, div: float, n_users: int, ld: int, d_W, d_bias, d_u, d_i, B: int, d_dS, d_X,
                         d_dPre, d_loss, stream=None, ordered=None):
    ws, nb = ordered.reserve(2 * B, ld) if ordered is not None else (0, 0)
    _check(load().qrec_buir_batch_loss_grad(_dp(d_S_on), _dp(d_S_tar), div, n_users, ld, _dp(d_W), _dp(d_bias), _dp(d_u),
                                            _dp(d_i), B, _dp(d_dS), _dp(d_X), _dp(d_dPre), _dp(d_loss), ws, nb, _sh(stream)))


def buir_wgrad_scratch_bytes(ld: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_buir_wgrad_scratch_bytes(ld, C.byref(out)))
    return out.value


def buir_wgrad(d_X, d_dPre, n_rows: int, ld: int, d_scratch, d_gW, d_gb, stream=None):
    _check(load().qrec_buir_wgrad(_dp(d_X), _dp(d_dPre), n_rows, ld, _dp(d_scratch), _dp(d_gW), _dp(d_gb), _sh(stream)))


def ema_update(d_target, d_online, tau: float, n_elems: int, stream=None):
    _check(load().qrec_ema_update(_dp(d_target), _dp(d_online), tau, n_elems, _sh(stream)))


def rank_hits(d_ids, n_batch_users: int, row_stride: int, n_cut: int, d_user_ids, d_test_indptr, d_test_items,
              d_discount, d_hits_out, d_dcg_out, stream=None):
    _check(load().qrec_rank_hits(_dp(d_ids), n_batch_users, row_stride, n_cut, _dp(d_user_ids), _dp(d_test_indptr),
                                 _dp(d_test_items), _dp(d_discount), _dp(d_hits_out), _dp(d_dcg_out), _sh(stream)))


def mark_batch_rows(d_u, d_i, d_j, B: int, n_users: int, d_row_mask, stream=None):
    _check(load().qrec_mark_batch_rows(_dp(d_u), _dp(d_i), _dp(d_j), B, n_users, _dp(d_row_mask), _sh(stream)))


def spmm_csr(plan, d_X, d_Y, ld: int, d_addend=None, addend_scale: float = 0.0, d_accum=None, stream=None,
             d_x_row_mask=None, d_y_row_mask=None, d_accum_init=None, d_addend_row_mask=None):
    """plan: qrec_amd.graph.SpmmPlan (device-resident segment arrays)"""
    _check(load().qrec_spmm_csr(_dp(plan.seg_row), _dp(plan.seg_beg), _dp(plan.seg_len), _dp(plan.seg_slot),
                                plan.n_segs, _dp(plan.long_row), _dp(plan.long_first), _dp(plan.long_count),
                                plan.n_long, _dp(plan.indices), _dp(plan.values), _dp(d_X), _dp(d_Y),
                                _dp(plan.partial), ld, _dp(d_addend), addend_scale, _dp(d_accum), _dp(d_accum_init), _dp(d_x_row_mask),
                                _dp(d_y_row_mask), _dp(d_addend_row_mask), _sh(stream)))


def subgraph_values(d_u, d_i, d_pos_ui, d_pos_iu, n_edges: int, n_users: int, n_items: int, d_row_of_nnz, d_indices, nnz: int,
                    d_dinv_table, max_deg: int, d_cnt, d_deg, d_values, d_keep_rows=None, n_keep: int = 0, d_drop_users=None,
                    n_drop_users: int = 0, d_drop_items=None, n_drop_items: int = 0, d_flags=None, stream=None):
    """value array of a sub-graph over the full graph's CSR structure (include/qrec_hip.h, csrc/augment.hip)"""
    _check(load().qrec_subgraph_values(_dp(d_u), _dp(d_i), _dp(d_pos_ui), _dp(d_pos_iu), n_edges, n_users, n_items, _dp(d_keep_rows), n_keep,
                                       _dp(d_drop_users), n_drop_users, _dp(d_drop_items), n_drop_items, _dp(d_row_of_nnz), _dp(d_indices), nnz,
                                       _dp(d_dinv_table), max_deg, _dp(d_cnt), _dp(d_deg), _dp(d_flags), _dp(d_values), _sh(stream)))


class OrderedScatter:
    """Workspace of the ordered (deterministic) gradient scatters -- the parity mode of qrec_bpr_batch_loss_grad,
    qrec_buir_batch_loss_grad and qrec_sept_ssl_loss_grad (include/qrec_hip.h, csrc/ordered.hip).  One per trainer: the calls
    that take ``ordered=`` run on the trainer's stream one after another and share it; grown on demand."""

    def __init__(self):
        self.buf = None

    def reserve(self, n_slots: int, ld: int):
        out = C.c_int64(0)
        _check(load().qrec_ordered_scatter_workspace_bytes(max(int(n_slots), 1), ld, C.byref(out)))
        if self.buf is None or self.buf.nbytes < out.value:
            device_sync()                   # a launch in flight may still be using the old workspace
            self.buf = DeviceBuffer(out.value, np.uint8)
        return self.buf.ptr, self.buf.nbytes


def scatter_add_rows_ordered(d_src, d_dst_rows, n_slots: int, ld: int, d_out, ordered: "OrderedScatter", class_size: int = 0, stream=None):
    """d_out[dst_rows[s]] += d_src[s], every row's slots in ascending s (class by class): np.add.at's order, bit for bit"""
    ws, nb = ordered.reserve(n_slots, ld)
    _check(load().qrec_scatter_add_rows_ordered(_dp(d_src), _dp(d_dst_rows), n_slots, ld, class_size, _dp(d_out), ws, nb, _sh(stream)))


def bpr_batch_loss_grad(d_S, div: float, n_users: int, n_rows: int, ld: int, d_u, d_i, d_j, B: int, eps: float,
                        reg: float, d_dE, d_loss, stream=None, d_row_mask=None, ordered=None):
    ws, nb = ordered.reserve(3 * B, ld) if ordered is not None else (0, 0)
    _check(load().qrec_bpr_batch_loss_grad(_dp(d_S), div, n_users, n_rows, ld, _dp(d_u), _dp(d_i), _dp(d_j), B,
                                           eps, reg, _dp(d_dE), _dp(d_loss), _dp(d_row_mask), ws, nb, _sh(stream)))


def adam_step(d_theta, d_m, d_v, d_grad, n_elems: int, grad_scale: float, alpha: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, stream=None, grad_l2: float = 0.0):
    _check(load().qrec_adam_step(_dp(d_theta), _dp(d_m), _dp(d_v), _dp(d_grad), n_elems, grad_scale, grad_l2, alpha,
                                 beta1, beta2, eps, _sh(stream)))


def perturb_rows(d_emb, n_rows: int, d: int, ld: int, eps: float, d_noise=None, seed: int = 0, stream_id: int = 0,
                 d_accum=None, stream=None, d_src=None, rows=None, philox_row0: int = 0):
    _check(load().qrec_perturb_rows(_dp(d_emb), _dp(d_src), n_rows, d, ld, eps, _dp(d_noise), seed & (2**64 - 1),
                                    stream_id & (2**64 - 1), _dp(d_accum), *_subset(rows), philox_row0, _sh(stream)))


def perturb_two_views(d_src, d_emb1, d_emb2, n_rows: int, d: int, ld: int, eps: float, d_noise1, d_noise2, seed: int,
                      stream_id1: int, stream_id2: int, d_sum1, d_sum2, d_src_sum, stream=None, rows=None, philox_row0: int = 0):
    """first layer of SimGCL's three encoders: both perturbed views of d_src, and the three layer sums START here"""
    m = 2**64 - 1
    _check(load().qrec_perturb_two_views(_dp(d_src), _dp(d_emb1), _dp(d_emb2), n_rows, d, ld, eps, _dp(d_noise1), _dp(d_noise2),
                                         seed & m, stream_id1 & m, stream_id2 & m, _dp(d_sum1), _dp(d_sum2), _dp(d_src_sum),
                                         *_subset(rows), philox_row0, _sh(stream)))


def info_nce_workspace_bytes(n: int, ld: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_info_nce_workspace_bytes(n, ld, C.byref(out)))
    return out.value


def info_nce_loss_grad(d_S1, d_S2, div: float, d_rows, n: int, ld: int, tau: float, cl_rate: float, d_workspace,
                       d_out, d_loss, stream=None, d_out2=None):
    _check(load().qrec_info_nce_loss_grad(_dp(d_S1), _dp(d_S2), div, _dp(d_rows), n, ld, tau, cl_rate,
                                          _dp(d_workspace), _dp(d_out), _dp(d_out2), _dp(d_loss), _sh(stream)))


def gate_fwd(d_X, d_W, d_bias, n_rows: int, ld: int, d_Y, d_S, stream=None):
    """Y = X * sigmoid(X W + b), S = the sigmoid (MHCN.py:109-112)"""
    _check(load().qrec_gate_fwd(_dp(d_X), _dp(d_W), _dp(d_bias), n_rows, ld, _dp(d_Y), _dp(d_S), _sh(stream)))


def gate_bwd(d_X, d_S, d_dY, d_W, n_rows: int, d: int, ld: int, d_Q, d_dX, accumulate: bool, dy_scale: float = 1.0, stream=None):
    _check(load().qrec_gate_bwd(_dp(d_X), _dp(d_S), _dp(d_dY), _dp(d_W), n_rows, d, ld, dy_scale, _dp(d_Q), _dp(d_dX),
                                1 if accumulate else 0, _sh(stream)))


def channel_attention_fwd(d_e, d_att, d_att_mat, d_half, n_rows: int, ld: int, d_v, d_score, d_out, stream=None):
    _check(load().qrec_channel_attention_fwd(_dp(d_e[0]), _dp(d_e[1]), _dp(d_e[2]), _dp(d_att), _dp(d_att_mat), _dp(d_half), n_rows, ld,
                                             _dp(d_v), _dp(d_score), _dp(d_out), _sh(stream)))


def channel_attention_bwd(d_dOut, d_e, d_score, d_v, d_att, d_att_mat, n_rows: int, ld: int, d_de, accumulate: bool, d_dhalf,
                          half_accumulate: bool, d_dv_scratch, d_g_att, d_g_att_mat, stream=None):
    _check(load().qrec_channel_attention_bwd(_dp(d_dOut), _dp(d_e[0]), _dp(d_e[1]), _dp(d_e[2]), _dp(d_score), _dp(d_v), _dp(d_att),
                                             _dp(d_att_mat), n_rows, ld, _dp(d_de[0]), _dp(d_de[1]), _dp(d_de[2]), 1 if accumulate else 0,
                                             _dp(d_dhalf), 1 if half_accumulate else 0, _dp(d_dv_scratch), _dp(d_g_att),
                                             _dp(d_g_att_mat), _sh(stream)))


def channel_attention_scratch_floats() -> int:
    out = C.c_int64(0)
    _check(load().qrec_channel_attention_scratch_bytes(C.byref(out)))
    return out.value // 4


def hss_scratch_bytes(n_rows: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_hss_scratch_bytes(n_rows, C.byref(out)))
    return out.value


def hss_loss_grad(d_em, d_edge, n_rows: int, d: int, ld: int, perms, scale: float, d_scratch, d_dem, d_dedge, d_loss, stream=None):
    """perms: device pointers (p1, p1inv, p2, p2inv, k2, k2inv, p3, p3inv, k3, k3inv) -- MHCN.py:184-206"""
    _check(load().qrec_hss_loss_grad(_dp(d_em), _dp(d_edge), n_rows, d, ld, *[_dp(p) for p in perms], scale, _dp(d_scratch),
                                     _dp(d_dem), _dp(d_dedge), _dp(d_loss), _sh(stream)))


def random_permutations_scratch_bytes(n: int, count: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_random_permutations_scratch_bytes(n, count, C.byref(out)))
    return out.value


def random_permutations(n: int, count: int, seed: int, stream_id: int, d_scratch, d_perms, d_invs=None, stream=None):
    """``count`` uniform shuffles of range(n) ([count][n]) with their inverses, one device sort"""
    _check(load().qrec_random_permutations(n, count, seed & (2**64 - 1), stream_id & (2**64 - 1), _dp(d_scratch), _dp(d_perms),
                                           _dp(d_invs), _sh(stream)))


def small_permutations(n: int, count: int, seed: int, stream_id: int, d_perms, d_invs, stream=None):
    _check(load().qrec_small_permutations(n, count, seed & (2**64 - 1), stream_id & (2**64 - 1), _dp(d_perms), _dp(d_invs), _sh(stream)))


def l2norm_rows_accum(d_X, n_rows: int, ld: int, d_S, d_inv, stream=None):
    """S[r] += l2_normalize(X[r]); inv[r] = 1 / max(|X[r]|, 1e-6)   (SEPT.py:144-160)"""
    _check(load().qrec_l2norm_rows_accum(_dp(d_X), n_rows, ld, _dp(d_S), _dp(d_inv), _sh(stream)))


def l2norm_rows_bwd(d_X, d_inv, d_dS, n_rows: int, ld: int, d_out, stream=None):
    _check(load().qrec_l2norm_rows_bwd(_dp(d_X), _dp(d_inv), _dp(d_dS), n_rows, ld, _dp(d_out), _sh(stream)))


def scale_copy(d_dst, d_src, n_elems: int, alpha: float, stream=None):
    _check(load().qrec_scale_copy(_dp(d_dst), _dp(d_src), n_elems, alpha, _sh(stream)))


def sept_ssl_workspace_bytes(n: int, ld: int, k: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_sept_ssl_workspace_bytes(n, ld, k, C.byref(out)))
    return out.value


def sept_ssl_loss_grad(d_S_friend, d_S_sharing, d_S_pref, d_S_aug, d_rows, n: int, ld: int, k: int, ss_rate: float, d_workspace,
                       d_dS_friend, d_dS_sharing, d_dS_pref, d_dS_aug, d_loss, d_labels=None, stream=None, ordered=None):
    """SEPT.py:214-262 on the batch's unique users (see include/qrec_hip.h)"""
    ws, nb = ordered.reserve(n * k, ld) if ordered is not None else (0, 0)
    _check(load().qrec_sept_ssl_loss_grad(_dp(d_S_friend), _dp(d_S_sharing), _dp(d_S_pref), _dp(d_S_aug), _dp(d_rows), n, ld, k,
                                          ss_rate, _dp(d_workspace), _dp(d_dS_friend), _dp(d_dS_sharing), _dp(d_dS_pref),
                                          _dp(d_dS_aug), _dp(d_loss), _dp(d_labels), ws, nb, _sh(stream)))


class RowSubset:
    """An ascending device-resident list of table rows (qrec_compact_marked_rows): the NGCF layer calls take one to
    restrict themselves to the batch's rows.  ``bound`` = the host's upper bound on the count (launch geometry)."""

    def __init__(self, capacity: int):
        self.rows = DeviceBuffer(max(capacity, 1), np.int32)
        self.count = DeviceBuffer.zeros(1, np.int32)
        self.capacity, self.bound = capacity, 0

    def from_mask(self, d_row_mask, n_rows: int, bound: int, stream=None):
        if bound > self.capacity:
            raise ValueError("RowSubset: bound above the capacity")
        _check(load().qrec_compact_marked_rows(_dp(d_row_mask), n_rows, _dp(self.rows), _dp(self.count), self.capacity, _sh(stream)))
        self.bound = bound
        return self


def mark_compact_batch_rows(d_u, d_i, d_j, B: int, n_users: int, n_rows: int, d_row_mask, rows=None, bound: int = 0, stream=None,
                            d_zero8=None, n_zero8: int = 0):
    """row bitmap of the batch (cleared first) + (``rows``: a RowSubset) its row list, one launch; ``d_zero8``: n_zero8 doubles
    cleared by the same launch (the step's loss accumulators)"""
    if rows is not None and bound > rows.capacity:
        raise ValueError("RowSubset: bound above the capacity")
    _check(load().qrec_mark_compact_batch_rows(_dp(d_u), _dp(d_i), _dp(d_j), B, n_users, n_rows, _dp(d_row_mask),
                                               _dp(rows.rows) if rows is not None else None, _dp(rows.count) if rows is not None else None,
                                               rows.capacity if rows is not None else 0, _dp(d_zero8), n_zero8, _sh(stream)))
    if rows is not None:
        rows.bound = bound
    return rows


def unique_per_batch(d_ids, n: int, batch: int, id_range: int, out_offset: int, d_rows, d_counts, stream=None):
    """distinct ids of every batch of an id stream, ascending, + out_offset: d_rows[b*batch ..), d_counts[b]"""
    _check(load().qrec_unique_per_batch(_dp(d_ids), n, batch, id_range, out_offset, _dp(d_rows), _dp(d_counts), _sh(stream)))


def _subset(rows):
    return (None, None, 0) if rows is None else (_dp(rows.rows), _dp(rows.count), rows.bound)


def ngcf_dense_fwd(d_E, d_side, d_W1, d_W2, n_rows: int, ld: int, d_pre, stream=None, rows: RowSubset | None = None):
    _check(load().qrec_ngcf_dense_fwd(_dp(d_E), _dp(d_side), _dp(d_W1), _dp(d_W2), n_rows, ld, _dp(d_pre), *_subset(rows), _sh(stream)))


def ngcf_activate(d_pre_gate, n_rows: int, d: int, ld: int, keep: float, d_mask, seed: int, stream_id: int, d_next,
                  d_wide, wide_ld: int, col_off: int, d_inv_norm, stream=None, rows: RowSubset | None = None, philox_row0: int = 0):
    _check(load().qrec_ngcf_activate(_dp(d_pre_gate), n_rows, d, ld, keep, _dp(d_mask), seed & (2**64 - 1),
                                     stream_id & (2**64 - 1), _dp(d_next), _dp(d_wide), wide_ld, col_off,
                                     _dp(d_inv_norm), *_subset(rows), philox_row0, _sh(stream)))


def ngcf_layer_bwd(d_dE_next, d_dWide, d_wide, wide_ld: int, col_off: int, d_inv_norm, d_gate, d_E, d_side, d_W1, d_W2,
                   n_rows: int, d: int, ld: int, d_dpre, d_dside, d_dE, d_partial, d_gW1, d_gW2, stream=None,
                   rows: RowSubset | None = None, d_wide_row_mask=None):
    _check(load().qrec_ngcf_layer_bwd(_dp(d_dE_next), _dp(d_dWide), _dp(d_wide), wide_ld, col_off, _dp(d_inv_norm),
                                      _dp(d_gate), _dp(d_E), _dp(d_side), _dp(d_W1), _dp(d_W2), n_rows, d, ld,
                                      _dp(d_dpre), _dp(d_dside), _dp(d_dE), _dp(d_partial), _dp(d_gW1), _dp(d_gW2),
                                      *_subset(rows), _dp(d_wide_row_mask), _sh(stream)))


def ngcf_wgrad_partial_bytes(n_rows: int, ld: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_ngcf_wgrad_partial_bytes(n_rows, ld, C.byref(out)))
    return out.value


def copy_cols(d_dst, dst_ld: int, d_src, src_ld: int, src_col_off: int, n_rows: int, d: int, accumulate: bool,
              stream=None, rows=None):
    _check(load().qrec_copy_cols(_dp(d_dst), dst_ld, _dp(d_src), src_ld, src_col_off, n_rows, d, 1 if accumulate else 0,
                                 *_subset(rows), _sh(stream)))


def zero_rows(d_X, ld: int, rows: RowSubset, stream=None):
    _check(load().qrec_zero_rows(_dp(d_X), ld, *_subset(rows), _sh(stream)))


def bpr_sgd_hogwild_item_major(d_P, d_Q, d: int, ld: int, d_u, d_i, d_j, n: int, chunk: int, grid_groups: int,
                               flush_every: int, lr: float, regU: float, regI: float, d_loss, stream=None,
                               d_driver_state=None, p_rows: int | None = None, q_rows: int | None = None, variant: int = HW_DEFAULT):
    """``variant``: HW_DEFAULT (atomic deltas on P[u] and Q[j]) or HW_P_RMW (P[u] by sc1 load + store: include/qrec_hip.h)"""
    _check(load().qrec_bpr_sgd_hogwild_item_major(_dp(d_P), _dp(d_Q), p_rows or _table_rows(d_P, ld), q_rows or _table_rows(d_Q, ld), d, ld, _dp(d_u), _dp(d_i), _dp(d_j), n, chunk,
                                                  grid_groups, flush_every, lr, regU, regI, _dp(d_loss), variant,
                                                  _dp(d_driver_state), _sh(stream)))


DRV_LR, DRV_LAST_LOSS, DRV_EPOCHS, DRV_CONVERGED, DRV_FAILED, DRV_WORDS = 0, 1, 2, 3, 4, 8
DRV_LOG_WORDS, STATS_WORDS = 8, 8 + 2 * 256


def epoch_close(d_P, p_rows: int, d_Q, q_rows: int, dtype: int, ld: int, d_stats, d_state, regU: float, regI: float,
                max_lr: float, tol: float, d_log=None, log_capacity: int = 0, stream=None):
    _check(load().qrec_epoch_close(_dp(d_P), p_rows, _dp(d_Q), q_rows, dtype, ld, _dp(d_stats), _dp(d_state), regU, regI,
                                   max_lr, tol, _dp(d_log), log_capacity, _sh(stream)))


def mt_sample_range(state625: np.ndarray, n: int, k: int) -> np.ndarray:
    """random.sample(range(n), k) with the exact CPython draw sequence"""
    from . import require_reference_python
    require_reference_python("qrec_mt_sample_range (random.sample's pool / selection-set switch)")
    _req(state625, np.uint32, "state625")
    out = np.empty(k, dtype=np.int64)
    _check(load().qrec_mt_sample_range(_hp(state625), n, k, _hp(out)))
    return out


# ---- multi-GPU: RCCL collectives and the kernels around them (include/qrec_hip.h, last section) -----------------
COMM_UID_BYTES = 128


def comm_library() -> tuple[str, int]:
    """(path of the librccl this process bound, its version code); raises when RCCL cannot be loaded"""
    path = C.create_string_buffer(1024); ver = C.c_int(0)
    _check(load().qrec_comm_library(path, 1024, C.byref(ver)))
    return path.value.decode(), ver.value


def comm_unique_id() -> bytes:
    uid = C.create_string_buffer(COMM_UID_BYTES)
    _check(load().qrec_comm_unique_id(uid))
    return uid.raw


class Comm:
    """One RCCL communicator (one process per GPU).  Every method only enqueues on ``stream``; buffers are DeviceBuffers
    or raw device addresses.  Row counts of ``alltoall_rows`` are host sequences of ``world`` entries."""

    def __init__(self, world: int, rank: int, uid: bytes, identity_shortcut: bool = True):
        """``identity_shortcut``: with a world of one an in-place all-reduce is the identity and is not handed to RCCL
        (measured: RCCL's one-rank path launches no kernel but costs ~90 us of stream time per grouped call), and the
        row exchanges are device-to-device copies of the rank's own segments; False sends everything through RCCL anyway
        (the binding's own test)."""
        ensure_init()
        if len(uid) != COMM_UID_BYTES:
            raise ValueError("comm uid must be 128 bytes")
        h = C.c_void_p()
        _check(load().qrec_comm_init(world, rank, C.create_string_buffer(uid, COMM_UID_BYTES), C.byref(h)))
        self.handle, self.world, self.rank = h.value, world, rank
        self._skip_identity = identity_shortcut and world == 1

    def allreduce(self, buf, count: int, dtype: int = F32, stream=None):
        if not self._skip_identity:
            _check(load().qrec_allreduce(self.handle, _dp(buf), count, dtype, _sh(stream)))

    def allreduce_pair(self, a, count_a: int, dtype_a: int, b, count_b: int, dtype_b: int, stream=None):
        if not self._skip_identity:
            _check(load().qrec_allreduce_pair(self.handle, _dp(a), count_a, dtype_a, _dp(b), count_b, dtype_b, _sh(stream)))

    def allgather(self, send, recv, count: int, dtype: int = F32, stream=None):
        _check(load().qrec_allgather(self.handle, _dp(send), _dp(recv), count, dtype, _sh(stream)))

    def reduce_scatter(self, send, recv, count: int, dtype: int = F32, stream=None):
        _check(load().qrec_reduce_scatter(self.handle, _dp(send), _dp(recv), count, dtype, _sh(stream)))

    def alltoall_rows(self, send, send_rows, recv, recv_rows, row_bytes: int, stream=None):
        s = np.ascontiguousarray(send_rows, dtype=np.int64); r = np.ascontiguousarray(recv_rows, dtype=np.int64)
        if s.size != self.world or r.size != self.world:
            raise ValueError("alltoall_rows: one row count per rank expected")
        if self._skip_identity:
            if int(s[0]) != int(r[0]):
                raise ValueError("alltoall_rows: a rank sends itself as many rows as it receives")
            if int(s[0]):
                memcpy_d2d(recv, send, int(s[0]) * row_bytes, stream)
            return
        _check(load().qrec_alltoall_rows(self.handle, _dp(send), _hp(s), _dp(recv), _hp(r), row_bytes, _sh(stream)))

    def sendrecv_segments(self, send, sends, recv, recvs, stream=None):
        """``sends`` / ``recvs``: lists of (peer, byte offset, bytes); one fused launch (qrec_sendrecv_segments)"""
        if self._skip_identity:
            if [b for _, _, b in sends] != [b for _, _, b in recvs]:
                raise ValueError("sendrecv_segments: a rank's segments to itself must match one by one")
            for (_, so, nb), (_, ro, _) in zip(sends, recvs):
                if nb:
                    memcpy_d2d(device_ptr(recv) + ro, device_ptr(send) + so, nb, stream)
            return
        sp = np.array([p for p, _, _ in sends], np.int32); so = np.array([o for _, o, _ in sends], np.int64); sb = np.array([b for _, _, b in sends], np.int64)
        rp = np.array([p for p, _, _ in recvs], np.int32); ro = np.array([o for _, o, _ in recvs], np.int64); rb = np.array([b for _, _, b in recvs], np.int64)
        _check(load().qrec_sendrecv_segments(self.handle, _dp(send), _hp(sp), _hp(so), _hp(sb), len(sends), _dp(recv), _hp(rp), _hp(ro), _hp(rb),
                                             len(recvs), _sh(stream)))

    def query(self) -> dict:
        """what RCCL reports: ranks in the communicator, this rank, the bound HIP device"""
        n, r, d = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        _check(load().qrec_comm_query(self.handle, C.byref(n), C.byref(r), C.byref(d)))
        return {"ranks": n.value, "rank": r.value, "device": d.value}

    def destroy(self):
        if getattr(self, "handle", None):
            load().qrec_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def dist_epoch_pre(d_P, p_rows: int, ld: int, d_Q, d_Q_start, d_delta, q_rows: int, d_stats, d_state=None, stream=None):
    _check(load().qrec_dist_epoch_pre(_dp(d_P), p_rows, ld, _dp(d_Q), _dp(d_Q_start), _dp(d_delta), q_rows, _dp(d_stats),
                                      _dp(d_state), _sh(stream)))


def dist_epoch_post(d_Q, d_Q_start, d_delta, q_rows: int, ld: int, d_stats, d_state, regU: float, regI: float, max_lr: float,
                    tol: float, d_log=None, log_capacity: int = 0, stream=None):
    _check(load().qrec_dist_epoch_post(_dp(d_Q), _dp(d_Q_start), _dp(d_delta), q_rows, ld, _dp(d_stats), _dp(d_state), regU, regI,
                                       max_lr, tol, _dp(d_log), log_capacity, _sh(stream)))


def table_delta(d_table, d_start, d_delta, n: int, stream=None):
    _check(load().qrec_table_delta(_dp(d_table), _dp(d_start), _dp(d_delta), n, _sh(stream)))


def table_apply(d_table, d_start, d_delta, n: int, stream=None):
    _check(load().qrec_table_apply(_dp(d_table), _dp(d_start), _dp(d_delta), n, _sh(stream)))


def shard_rows(n_items: int, world: int, rank: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_shard_rows(n_items, world, rank, C.byref(out)))
    return out.value


def shard_plan_scratch_bytes(n_items: int, world: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_shard_plan_scratch_bytes(n_items, world, C.byref(out)))
    return out.value


def shard_plan_batch(d_i, d_j, n: int, n_items: int, world: int, d_scratch, d_req_rows, d_counts, d_ci, d_cj, stream=None):
    _check(load().qrec_shard_plan_batch(_dp(d_i), _dp(d_j), n, n_items, world, _dp(d_scratch), _dp(d_req_rows), _dp(d_counts),
                                        _dp(d_ci), _dp(d_cj), _sh(stream)))


def shard_plan_epoch_scratch_bytes(n_items: int, world: int, n_batches: int) -> int:
    out = C.c_int64(0)
    _check(load().qrec_shard_plan_epoch_scratch_bytes(n_items, world, n_batches, C.byref(out)))
    return out.value


def shard_plan_epoch(d_i, d_j, d_bounds, n_batches: int, n: int, n_items: int, world: int, d_scratch, d_req_rows, d_req_off, d_counts,
                     d_ci, d_cj, stream=None):
    _check(load().qrec_shard_plan_epoch(_dp(d_i), _dp(d_j), _dp(d_bounds), n_batches, n, n_items, world, _dp(d_scratch), _dp(d_req_rows),
                                        _dp(d_req_off), _dp(d_counts), _dp(d_ci), _dp(d_cj), _sh(stream)))


def gather_rows(d_table, ld: int, d_rows, n: int, d_out, stream=None):
    _check(load().qrec_gather_rows(_dp(d_table), ld, _dp(d_rows), n, _dp(d_out), _sh(stream)))


def rows_gather_owned(d_block, ld: int, lo: int, hi: int, d_ids, n: int, d_out, stream=None):
    """d_out[k] = row d_ids[k] of a table whose rows [lo, hi) are d_block; rows held elsewhere: zeros"""
    _check(load().qrec_rows_gather_owned(_dp(d_block), ld, lo, hi, _dp(d_ids), n, _dp(d_out), _sh(stream)))


def rows_scatter_add_owned(d_block, ld: int, lo: int, hi: int, d_ids, n: int, d_src, stream=None):
    _check(load().qrec_rows_scatter_add_owned(_dp(d_block), ld, lo, hi, _dp(d_ids), n, _dp(d_src), _sh(stream)))


def batch_rows_gather(d_block, ld: int, lo: int, hi: int, d_u, d_i, d_j, B: int, n_users: int, d_out, stream=None):
    """d_out[3B][ld] = the batch's rows {u, n_users + i, n_users + j} of a table whose rows [lo, hi) are d_block; rows held elsewhere: zeros"""
    _check(load().qrec_batch_rows_gather(_dp(d_block), ld, lo, hi, _dp(d_u), _dp(d_i), _dp(d_j), B, n_users, _dp(d_out), _sh(stream)))


def batch_rows_scatter_add(d_block, ld: int, lo: int, hi: int, d_u, d_i, d_j, B: int, n_users: int, d_src, stream=None):
    """d_block[row - lo] += d_src[k] for the batch rows that lie in [lo, hi)"""
    _check(load().qrec_batch_rows_scatter_add(_dp(d_block), ld, lo, hi, _dp(d_u), _dp(d_i), _dp(d_j), B, n_users, _dp(d_src), _sh(stream)))


def scatter_add_row_deltas(d_table, ld: int, d_rows, n: int, d_fresh, d_sent, stream=None):
    _check(load().qrec_scatter_add_row_deltas(_dp(d_table), ld, _dp(d_rows), n, _dp(d_fresh), _dp(d_sent), _sh(stream)))
