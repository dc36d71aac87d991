"""ctypes binding of the plain-C oracle (oracle/qrec_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqrec_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "qrec_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32, i64, u32, u64, f64, f32 = (C.c_void_p, C.c_int32, C.c_int64, C.c_uint32,
                                            C.c_uint64, C.c_double, C.c_float)
        L.orc_seed_cpython.argtypes = [vp, u64]
        L.orc_seed_numpy.argtypes = [vp, u32]
        L.orc_set_state.argtypes = [vp, vp]
        L.orc_get_state.argtypes = [vp, vp]
        L.orc_state_size.restype = C.c_int
        L.orc_u32.argtypes = [vp]; L.orc_u32.restype = u32
        L.orc_random.argtypes = [vp]; L.orc_random.restype = f64
        L.orc_numpy_rand.argtypes = [vp, vp, i64]
        L.orc_randbelow.argtypes = [vp, u32]; L.orc_randbelow.restype = u32
        L.orc_shuffle.argtypes = [vp, vp, i64]
        L.orc_data_split.argtypes = [vp, i64, f64, vp]
        L.orc_sample_range.argtypes = [vp, i64, i64, vp]
        L.orc_bpr_sample_epoch.argtypes = [vp, vp, vp, i32, i32, vp]
        L.orc_bpr_sample_epoch.restype = i64
        L.orc_pairwise_sample_epoch.argtypes = [vp, vp, i64, vp, vp, i32, vp]
        L.orc_bpr_sgd_f64.argtypes = [vp, vp, i32, vp, vp, vp, i64, f64, f64, f64]
        L.orc_bpr_sgd_f64.restype = f64
        L.orc_bpr_sgd_f32.argtypes = [vp, vp, i32, vp, vp, vp, i64, f32, f32, f32]
        L.orc_bpr_sgd_f32.restype = f64
        L.orc_sumsq_f64.argtypes = [vp, i64]; L.orc_sumsq_f64.restype = f64
        L.orc_mf_sgd_f64.argtypes = [vp, vp, i32, vp, vp, vp, i64, f64]
        L.orc_mf_sgd_f64.restype = f64
        L.orc_mf_sgd_var_f64.argtypes = [C.c_int, vp, vp, vp, vp, i32, vp, vp, vp, i64, f64, f64, f64, f64, f64]
        L.orc_mf_sgd_var_f64.restype = f64
        L.orc_svdpp_sgd_f64.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, f64]
        L.orc_svdpp_sgd_f64.restype = f64
        L.orc_find_k_largest.argtypes = [i32, vp, i32, vp, vp]
        L.orc_find_k_largest.restype = C.c_int
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_philox_bpr_sample.argtypes = [vp, vp, vp, i64, i32, u64, u64, vp]
        L.orc_philox_perm_keys.argtypes = [i64, u64, u64, vp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _chk(a, dtype):
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous, (
        f"need C-contiguous {dtype}, got {getattr(a, 'dtype', type(a))}")
    return a


class MT:
    """CPython/numpy-compatible MT19937 stream."""

    def __init__(self):
        self._buf = np.zeros(lib().orc_state_size() // 4 + 1, dtype=np.uint32)

    @property
    def ptr(self):
        return _p(self._buf)

    @classmethod
    def cpython_seed(cls, a: int) -> "MT":
        m = cls(); lib().orc_seed_cpython(m.ptr, abs(int(a))); return m

    @classmethod
    def numpy_seed(cls, s: int) -> "MT":
        m = cls(); lib().orc_seed_numpy(m.ptr, int(s)); return m

    @classmethod
    def from_python_state(cls, state) -> "MT":
        """state = random.getstate()"""
        m = cls()
        w = np.array(state[1], dtype=np.uint32)
        lib().orc_set_state(m.ptr, _p(w))
        return m

    def python_state(self):
        w = np.zeros(625, dtype=np.uint32)
        lib().orc_get_state(self.ptr, _p(w))
        return (3, tuple(int(x) for x in w), None)

    def words625(self) -> np.ndarray:
        w = np.zeros(625, dtype=np.uint32)
        lib().orc_get_state(self.ptr, _p(w))
        return w

    def u32(self) -> int:
        return int(lib().orc_u32(self.ptr))

    def random(self) -> float:
        return float(lib().orc_random(self.ptr))

    def randbelow(self, n: int) -> int:
        return int(lib().orc_randbelow(self.ptr, n))

    def numpy_rand(self, *shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float64)
        lib().orc_numpy_rand(self.ptr, _p(out), out.size)
        return out

    def shuffle(self, n: int, perm: np.ndarray | None = None):
        if perm is not None:
            _chk(perm, np.int64); assert perm.size == n
        lib().orc_shuffle(self.ptr, _p(perm) if perm is not None else None, n)
        return perm

    def sample_range(self, n: int, k: int) -> np.ndarray:
        """random.sample(range(n), k)"""
        out = np.empty(k, dtype=np.int64)
        lib().orc_sample_range(self.ptr, n, k, _p(out))
        return out

    def data_split(self, n: int, ratio: float) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint8)
        lib().orc_data_split(self.ptr, n, ratio, _p(out))
        return out.astype(bool)


def bpr_sample_epoch(mt: MT, pos_indptr, pos_indices, n_items: int) -> np.ndarray:
    """model/ranking/BPR.py:28-38 -- one negative per CSR entry."""
    _chk(pos_indptr, np.int64); _chk(pos_indices, np.int32)
    j = np.empty(pos_indices.size, dtype=np.int32)
    lib().orc_bpr_sample_epoch(mt.ptr, _p(pos_indptr), _p(pos_indices),
                               pos_indptr.size - 1, n_items, _p(j))
    return j


def pairwise_sample_epoch(mt: MT, row_user, rated_indptr, rated_sorted, n_items: int):
    """base/deepRecommender.py:41-49 on already-shuffled rows."""
    _chk(row_user, np.int32); _chk(rated_indptr, np.int64); _chk(rated_sorted, np.int32)
    neg = np.empty(row_user.size, dtype=np.int32)
    lib().orc_pairwise_sample_epoch(mt.ptr, _p(row_user), row_user.size, _p(rated_indptr),
                                    _p(rated_sorted), n_items, _p(neg))
    return neg


def bpr_sgd(P, Q, u, i, j, lr, regU, regI) -> float:
    """model/ranking/BPR.py:45-53 over n triplets, in place.  dtype of P picks f64/f32."""
    _chk(u, np.int32); _chk(i, np.int32); _chk(j, np.int32)
    d = P.shape[1]
    if P.dtype == np.float64:
        _chk(P, np.float64); _chk(Q, np.float64)
        return lib().orc_bpr_sgd_f64(_p(P), _p(Q), d, _p(u), _p(i), _p(j), u.size, lr, regU, regI)
    _chk(P, np.float32); _chk(Q, np.float32)
    return lib().orc_bpr_sgd_f32(_p(P), _p(Q), d, _p(u), _p(i), _p(j), u.size, lr, regU, regI)


def tbpr_sample_epoch(mt: MT, pos_indptr, pos_items, n_items: int, joint, weak, strong):
    """model/ranking/TBPR.py:131-158 -- the epoch's chained (u, a, b) triplets; joint/weak/strong = (indptr, items)."""
    _chk(pos_indptr, np.int64); _chk(pos_items, np.int32)
    for ptr, items in (joint, weak, strong):
        _chk(ptr, np.int64); _chk(items, np.int32)
    cap = 4 * pos_items.size
    u, a, b = (np.empty(cap, dtype=np.int32) for _ in range(3))
    L = lib()
    L.orc_tbpr_sample_epoch.restype = C.c_int64
    L.orc_tbpr_sample_epoch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 9
    n = L.orc_tbpr_sample_epoch(mt.ptr, _p(pos_indptr), _p(pos_items), pos_indptr.size - 1, n_items, _p(joint[0]), _p(joint[1]),
                                _p(weak[0]), _p(weak[1]), _p(strong[0]), _p(strong[1]), _p(u), _p(a), _p(b))
    return u[:n].copy(), a[:n].copy(), b[:n].copy()


def tbpr_epoch(P, Q, u, a, b, lr, regU, regI) -> float:
    """TBPR.optimization over the chained triplets + the per-user regularisation terms of the loss (TBPR.py:157-159)"""
    _chk(P, np.float64); _chk(Q, np.float64); _chk(u, np.int32); _chk(a, np.int32); _chk(b, np.int32)
    L = lib()
    L.orc_tbpr_epoch_f64.restype = C.c_double
    L.orc_tbpr_epoch_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_double, C.c_double, C.c_double]
    return L.orc_tbpr_epoch_f64(_p(P), _p(Q), P.shape[1], P.shape[0], Q.shape[0], _p(u), _p(a), _p(b), u.size, lr, regU, regI)


def sumsq(x) -> float:
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().orc_sumsq_f64(_p(x), x.size)


def mf_sgd(P, Q, u, i, r, lr) -> float:
    """model/rating/BasicMF.py:9-26 over n ratings, in place (fp64)."""
    _chk(P, np.float64); _chk(Q, np.float64); _chk(u, np.int32); _chk(i, np.int32)
    _chk(r, np.float64)
    return lib().orc_mf_sgd_f64(_p(P), _p(Q), P.shape[1], _p(u), _p(i), _p(r), u.size, lr)


def mf_sgd_variant(variant: int, P, Q, u, i, r, lr, regU=0.0, regI=0.0, Bu=None, Bi=None, regB=0.0, gmean=0.0) -> float:
    """variant 1 = model/rating/PMF.py:9-28, 2 = model/rating/SVD.py:13-35, 3 = model/rating/EE.py:15-34 (fp64, in place)."""
    _chk(P, np.float64); _chk(Q, np.float64); _chk(u, np.int32); _chk(i, np.int32); _chk(r, np.float64)
    if variant in (2, 3):
        _chk(Bu, np.float64); _chk(Bi, np.float64)
    return lib().orc_mf_sgd_var_f64(variant, _p(P), _p(Q), _p(Bu) if Bu is not None else None,
                                    _p(Bi) if Bi is not None else None, P.shape[1], _p(u), _p(i), _p(r), u.size,
                                    lr, regU, regI, regB, gmean)


def svdpp_sgd(P, Q, Y, Bu, Bi, rated_indptr, rated_items, u, i, r, lr, regU, regI, regB, regY, gmean) -> float:
    """model/rating/SVDPlusPlus.py:25-62 over n ratings (fp64, in place); rated_*: data.userRated order as CSR."""
    for a in (P, Q, Y, Bu, Bi, r):
        _chk(a, np.float64)
    _chk(rated_indptr, np.int64); _chk(rated_items, np.int32); _chk(u, np.int32); _chk(i, np.int32)
    return lib().orc_svdpp_sgd_f64(_p(P), _p(Q), _p(Y), _p(Bu), _p(Bi), P.shape[1], _p(rated_indptr), _p(rated_items),
                                   _p(u), _p(i), _p(r), u.size, lr, regU, regI, regB, regY, gmean)


def find_k_largest(K: int, cand):
    """util/qmath.py:134-146 with CPython heapq tie behaviour."""
    cand = np.ascontiguousarray(cand, dtype=np.float64)
    k = min(K, cand.size)
    ids = np.empty(k, dtype=np.int32); sc = np.empty(k, dtype=np.float64)
    lib().orc_find_k_largest(K, _p(cand), cand.size, _p(ids), _p(sc))
    return ids, sc


def philox4x32_10(ctr, key) -> np.ndarray:
    """one Philox4x32-10 block (Random123): 4 counter words, 2 key words -> 4 output words"""
    c, k, out = np.asarray(ctr, dtype=np.uint32).copy(), np.asarray(key, dtype=np.uint32).copy(), np.zeros(4, dtype=np.uint32)
    assert c.size == 4 and k.size == 2
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def philox_permutation(n: int, seed: int, stream_id: int) -> np.ndarray:
    """the device's uniform permutation of range(n) (qrec_random_permutations, count = 1) on the CPU: stable argsort of the Philox keys"""
    keys = np.empty(n, dtype=np.uint64)
    lib().orc_philox_perm_keys(n, int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFFFFFFFFFF, _p(keys))
    return np.argsort(keys, kind="stable").astype(np.int32)


def philox_bpr_sample(indptr, sorted_items, row_user, n_items: int, seed: int, epoch: int) -> np.ndarray:
    """the throughput-mode negative sampler (qrec_philox_bpr_sample) on the CPU: j[t] for every stored position t"""
    _chk(indptr, np.int64); _chk(sorted_items, np.int32); _chk(row_user, np.int32)
    out = np.empty(row_user.size, dtype=np.int32)
    lib().orc_philox_bpr_sample(_p(indptr), _p(sorted_items), _p(row_user), row_user.size, n_items,
                                int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFFFFFFFFFF, _p(out))
    return out
