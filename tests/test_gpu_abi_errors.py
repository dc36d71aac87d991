"""Error behaviour and degenerate inputs of the C ABI (include/qrec_hip.h): every entry point returns a negative
code with a message in qrec_last_error() for arguments it cannot serve -- the reference prints and exits for the
equivalent situations (util/config.py:8-10, base/iterativeRecommender.py:84-86) -- and treats empty inputs as no-ops."""
import ctypes as C

import numpy as np
import pytest

from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def _err(call):
    with pytest.raises(capi.QRecError) as e:
        call()
    assert e.value.code < 0 and len(str(e.value)) > 20       # a message, not just a code
    return str(e.value)


def test_null_and_bad_arguments_are_refused_with_a_message():
    L = capi.load()
    f32 = lambda *shape: DB.zeros(shape, np.float32)
    i32 = lambda n: DB.zeros(n, np.int32)
    T, idx, out = f32(100, 64), i32(16), DB.zeros(8, np.float64)
    assert "null" in _err(lambda: capi.bpr_sgd_ordered(None, T, capi.F32, 64, 64, idx, idx, idx, 16, 0.1, 0, 0, out))
    assert "dtype" in _err(lambda: capi.bpr_sgd_ordered(T, T, 7, 64, 64, idx, idx, idx, 16, 0.1, 0, 0, out))
    assert "ld" in _err(lambda: capi.bpr_sgd_ordered(T, T, capi.F32, 64, 32, idx, idx, idx, 16, 0.1, 0, 0, out))
    _err(lambda: capi.bpr_sgd_hogwild(T, T, 64, 64, idx, idx, idx, 16, 0, 0, 0.1, 0, 0, out))            # chunk 0
    _err(lambda: capi.bpr_sgd_hogwild(T, T, 64, 64, idx, idx, idx, 16, 10 ** 6, 0, 0.1, 0, 0, out))      # chunk too large
    _err(lambda: capi.bpr_sgd_hogwild_item_major(T, T, 64, 64, idx, idx, idx, 16, 32, 0, 0, 0.1, 0, 0, out))   # flush 0
    _err(lambda: capi.mf_sgd_ordered(T, T, capi.F32, 64, 64, idx, idx, out, 16, 0.1, out, variant=5))
    _err(lambda: capi.mf_sgd_ordered(T, T, capi.F32, 64, 64, idx, idx, out, 16, 0.1, out, variant=capi.MF_SVD))  # no biases
    _err(lambda: capi.sumsq(T, 9, 100, 64, 64, out))
    _err(lambda: capi.adam_step(None, T, T, T, 6400, 1.0, 0.001))
    _err(lambda: capi.bpr_batch_loss_grad(T, 0.0, 50, 100, 64, idx, idx, idx, 16, 1e-7, 0.0, T, out))       # div = 0
    _err(lambda: capi.bpr_batch_loss_grad(T, 1.0, 50, 100, 48, idx, idx, idx, 16, 1e-7, 0.0, T, out))       # stride 48
    _err(lambda: capi.score_topk(T, T, capi.F32, 64, 64, 100, idx, 16, None, None, 0, T, idx, T))           # N = 0
    _err(lambda: capi.score_topk(T, T, capi.F32, 64, 64, 100, idx, 16, None, None, 101, T, idx, T))         # N > 100
    _err(lambda: capi.score_topk(T, T, capi.F32, 50, 50, 100, idx, 16, None, None, 10, T, idx, T))          # unpadded stride
    _err(lambda: capi.rank_hits(idx, 4, 2, 3, idx, DB.zeros(8, np.int64), idx, out, idx, out))              # stride < cut
    _err(lambda: capi.epoch_close(T, 100, T, 100, capi.F32, 63, out, out, 0, 0, 1, 1e-3))                   # stride not x4
    _err(lambda: capi.info_nce_loss_grad(T, T, 0.0, idx, 8, 64, 0.2, 0.5, T, T, out))                        # div = 0
    _err(lambda: capi.ngcf_dense_fwd(T, T, T, T, 100, 48, T))
    _err(lambda: capi.buir_batch_loss_grad(T, T, 3.0, 50, 256, T, T, idx, idx, 8, T, T, T, out))            # stride 256
    _err(lambda: capi.ema_update(T, T, 0.9, 6401))                                                         # not x4
    code = L.qrec_memcpy_d2h(None, C.c_void_p(T.ptr), C.c_int64(16), None)
    assert code < 0 and L.qrec_last_error()


def test_in_place_spmm_is_refused_and_plan_must_be_complete():
    import scipy.sparse as sp
    from qrec_amd.graph import SpmmPlan
    A = sp.random(200, 200, density=0.05, format="csr", dtype=np.float32, random_state=1)
    plan = SpmmPlan(A.indptr.astype(np.int64), A.indices, A.data, 64)
    X = DB.zeros((200, 64), np.float32)
    assert "in-place" in _err(lambda: capi.spmm_csr(plan, X, X, 64))
    _err(lambda: capi.spmm_csr(plan, X, DB.zeros((200, 64), np.float32), 48))


def test_empty_inputs_are_no_ops():
    T0 = np.random.default_rng(0).random((50, 64)).astype(np.float32)
    P, Q = DB.from_numpy(T0), DB.from_numpy(T0)
    idx, out = DB.zeros(1, np.int32), DB.from_numpy(np.array([7.0, 0, 0, 0], np.float64))
    capi.bpr_sgd_hogwild(P, Q, 64, 64, idx, idx, idx, 0, 32, 0, 0.1, 0.01, 0.01, out)
    capi.bpr_sgd_hogwild_item_major(P, Q, 64, 64, idx, idx, idx, 0, 32, 0, 8, 0.1, 0.01, 0.01, out)
    capi.bpr_batch_loss_grad(P, 1.0, 25, 50, 64, idx, idx, idx, 0, 1e-7, 0.0, Q, out)
    capi.mark_batch_rows(idx, idx, idx, 0, 25, DB.zeros(2, np.uint32))
    capi.buir_batch_loss_grad(P, Q, 3.0, 25, 64, P, Q, idx, idx, 0, P, Q, Q, out)
    assert np.array_equal(P.numpy(), T0) and np.array_equal(Q.numpy(), T0) and out.numpy()[0] == 7.0
    capi.bpr_sgd_ordered(P, Q, capi.F32, 64, 64, idx, idx, idx, 0, 0.1, 0, 0, out)        # n = 0: loss := 0
    assert out.numpy()[0] == 0.0 and np.array_equal(P.numpy(), T0)
    # empty rating file / empty split
    from qrec_amd.data.rows import RatingRows
    rows = RatingRows(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), [], [])
    assert len(rows) == 0 and rows.to_list() == [] and len(rows[0:0]) == 0
