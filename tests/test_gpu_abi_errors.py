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


def test_social_and_graph_model_entry_points_refuse_bad_arguments_and_accept_empty_ones():
    """The entry points that came in with SVD++, TBPR, SEPT and MHCN: same contract -- negative code + message for what
    they cannot serve, empty inputs are no-ops that leave tables and accumulated losses alone."""
    f32 = lambda *shape: DB.zeros(shape, np.float32)
    T0 = np.random.default_rng(1).random((60, 64)).astype(np.float32)
    T, S, idx = DB.from_numpy(T0), DB.from_numpy(T0), DB.zeros(16, np.int32)
    out = DB.from_numpy(np.array([5.0, 6.0, 0, 0, 0, 0, 0, 0], np.float64))
    ws = DB.zeros(1 << 20, np.uint8)
    # bad arguments
    assert "stride" in _err(lambda: capi.l2norm_rows_accum(T, 60, 48, S, f32(60)))
    assert "stride" in _err(lambda: capi.l2norm_rows_bwd(T, f32(60), T, 60, 48, S))
    _err(lambda: capi.scale_copy(S, T, 6401, 0.5))                                                    # not a multiple of 4
    assert "ins_cnt" in _err(lambda: capi.sept_ssl_loss_grad(T, T, T, T, idx, 4, 64, 9, 0.1, ws, S, S, S, S, out))   # k > unique users
    _err(lambda: capi.sept_ssl_loss_grad(T, T, T, T, None, 4, 64, 2, 0.1, ws, S, S, S, S, out))        # no row list
    _err(lambda: capi.sept_ssl_loss_grad(T, T, T, T, idx, 4, 48, 2, 0.1, ws, S, S, S, S, out))         # stride 48
    _err(lambda: capi.gate_fwd(T, None, f32(64), 60, 64, S, S))
    _err(lambda: capi.gate_fwd(T, f32(48, 48), f32(48), 60, 48, S, S))
    _err(lambda: capi.gate_bwd(T, T, T, f32(64, 64), 60, 65, 64, S, S, False))                          # d > ld
    _err(lambda: capi.hss_loss_grad(T, T, 60, 70, 64, [idx] * 10, 1.0, ws, S, S, out))                 # d > ld
    _err(lambda: capi.random_permutations(1 << 20, 1 << 12, 0, 0, ws, idx))                            # n * count >= 2^31
    _err(lambda: capi.small_permutations(5000, 1, 0, 0, idx, idx))                                     # n > 4096
    _err(lambda: capi.tbpr_sgd_ordered(T, T, 9, 64, 64, idx, idx, idx, 16, 0.1, 0, 0, out, out))        # dtype
    _err(lambda: capi.tbpr_sgd_ordered(T, T, capi.F32, 64, 64, idx, idx, idx, 16, 0.1, 0, 0, None, out))  # no sums
    _err(lambda: capi.svdpp_sgd_ordered(T, T, None, T, T, capi.F32, 64, 64, DB.zeros(61, np.int64), idx, idx, idx, out, 4, 0.1, 0, 0, 0, 0, 3.0, out))
    _err(lambda: capi.sbpr_sgd_ordered(T, T, T, 9, 64, 64, idx, 3, 0.1, 0, 0, 0.0, out, out))               # dtype
    _err(lambda: capi.sbpr_sgd_ordered(T, T, None, capi.F32, 64, 64, idx, 3, 0.1, 0, 0, 0.0, out, out))     # no biases
    _err(lambda: capi.sbpr_sgd_ordered(T, T, T, capi.F32, 64, 64, None, 3, 0.1, 0, 0, 0.0, out, out))        # rows missing
    assert np.array_equal(T.numpy(), T0) and np.array_equal(S.numpy(), T0)
    # empty inputs
    capi.l2norm_rows_accum(T, 0, 64, S, f32(1)); capi.l2norm_rows_bwd(T, f32(1), T, 0, 64, S); capi.scale_copy(S, T, 0, 0.5)
    capi.sept_ssl_loss_grad(T, T, T, T, None, 0, 64, 2, 0.1, ws, S, S, S, S, out)
    capi.gate_fwd(T, f32(64, 64), f32(64), 0, 64, S, S); capi.gate_bwd(T, T, T, f32(64, 64), 0, 64, 64, S, S, True)
    capi.hss_loss_grad(T, T, 0, 64, 64, [idx] * 10, 1.0, ws, S, S, out)
    capi.random_permutations(0, 3, 0, 0, ws, idx); capi.small_permutations(8, 0, 0, 0, idx, idx)
    assert np.array_equal(S.numpy(), T0) and out.numpy()[0] == 5.0 and out.numpy()[1] == 6.0
    capi.tbpr_sgd_ordered(T, T, capi.F32, 64, 64, idx, idx, idx, 0, 0.1, 0.01, 0.01, out, out)          # n = 0: both loss terms := 0
    assert out.numpy()[0] == 0.0 and out.numpy()[1] == 0.0 and np.array_equal(T.numpy(), T0)
    u, a, b = capi.mt_tbpr_sample_epoch(np.zeros(625, np.uint32), np.zeros(4, np.int64), np.zeros(0, np.int32), 10,
                                        *[(np.zeros(4, np.int64), np.zeros(0, np.int32))] * 3)
    assert u.size == a.size == b.size == 0
    capi.sbpr_sgd_ordered(T, T, T, capi.F32, 64, 64, None, 0, 0.1, 0.01, 0.01, 2.0, out, out)               # n = 0: no user visited, both loss terms := 0
    assert out.numpy()[0] == 0.0 and out.numpy()[1] == 0.0 and np.array_equal(T.numpy(), T0)
    rows = capi.mt_sbpr_sample_epoch(np.zeros(625, np.uint32), np.zeros(0, np.int32), np.zeros(4, np.int64), np.zeros(0, np.int32), 10,
                                     np.zeros(4, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), np.full(10, -1, np.int32), np.zeros(3, np.uint8))
    assert rows.shape == (0, 5)


def test_round_3_entry_points_refuse_what_they_cannot_serve():
    """the item-major kernel's update policies, the exact mode's wide layout, page-locked memory and the grouped send/recv"""
    L = capi.load()
    T, idx, out = DB.zeros((100, 64), np.float32), DB.zeros(16, np.int32), DB.zeros(8, np.float64)
    _err(lambda: capi.bpr_sgd_hogwild_item_major(T, T, 64, 64, idx, idx, idx, 16, 32, 0, 16, 0.1, 0, 0, out, variant=99))          # unknown update policy
    _err(lambda: capi.bpr_sgd_hogwild_item_major(T, T, 64, 64, idx, idx, idx, 16, 32, 0, 16, 0.1, 0, 0, out, variant=capi.HW_SC1_RMW))   # a user-major policy
    capi.bpr_sgd_hogwild_item_major(T, T, 64, 64, idx, idx, idx, 0, 32, 0, 16, 0.1, 0, 0, out, variant=capi.HW_P_RMW)               # an empty epoch is a no-op
    assert not T.numpy().any()
    # page-locked memory
    h = C.c_void_p()
    assert L.qrec_host_alloc(C.c_int64(0), C.byref(h)) < 0 and L.qrec_last_error()
    assert L.qrec_host_alloc(C.c_int64(64), None) < 0
    assert L.qrec_memcpy_d2h_async(None, C.c_void_p(T.ptr), C.c_int64(16), None) < 0
    pin = capi.PinnedBuffer(4, np.int32)
    with pytest.raises(ValueError):
        capi.memcpy_d2h_async(pin, idx, 64)                                               # more than the buffer holds
    src = DB.from_numpy(np.arange(4, dtype=np.int32)); ev = capi.Event()
    capi.memcpy_d2h_async(pin, src, 16); ev.record(); ev.sync()
    assert pin.a.tolist() == [0, 1, 2, 3]


def test_round4_entry_points_refuse_bad_arguments_and_treat_empty_inputs_as_no_ops():
    """qrec_batch_rows_gather / _scatter_add and qrec_rows_gather_owned / _scatter_add_owned (a batch's rows out of / into row-partitioned tables)."""
    f32 = lambda *shape: DB.zeros(shape, np.float32)
    T, S, idx, out = f32(100, 64), f32(100, 64), DB.from_numpy(np.arange(16, dtype=np.int32)), f32(48, 64)
    for call in (lambda ld: capi.batch_rows_gather(T, ld, 0, 100, idx, idx, idx, 16, 50, out),
                 lambda ld: capi.batch_rows_scatter_add(T, ld, 0, 100, idx, idx, idx, 16, 50, out),
                 lambda ld: capi.rows_gather_owned(T, ld, 0, 100, idx, 16, out),
                 lambda ld: capi.rows_scatter_add_owned(T, ld, 0, 100, idx, 16, out)):
        assert "stride" in _err(lambda: call(48))                         # a row stride the kernels have no lane mapping for
    assert "null" in _err(lambda: capi.batch_rows_gather(T, 64, 0, 100, idx, None, idx, 16, 50, out))
    assert "null" in _err(lambda: capi.rows_gather_owned(T, 64, 0, 100, idx, 16, None))
    _err(lambda: capi.batch_rows_gather(T, 64, 10, 5, idx, idx, idx, 16, 50, out))      # hi < lo
    _err(lambda: capi.rows_scatter_add_owned(T, 64, -1, 100, idx, 16, out))             # lo < 0
    # empty inputs: nothing is touched, nothing is dereferenced
    before = T.numpy().copy()
    capi.batch_rows_gather(None, 64, 0, 0, None, None, None, 0, 0, None); capi.batch_rows_scatter_add(None, 64, 0, 0, None, None, None, 0, 0, None)
    capi.rows_gather_owned(None, 64, 0, 0, None, 0, None); capi.rows_scatter_add_owned(None, 64, 0, 0, None, 0, None)
    capi.device_sync()
    assert np.array_equal(T.numpy(), before)
    # a block that holds none of the batch's rows: gather gives zeros, scatter adds nothing
    ones = DB.from_numpy(np.ones((48, 64), np.float32))
    capi.batch_rows_gather(T, 64, 5000, 5100, idx, idx, idx, 16, 50, ones)
    assert not ones.numpy().any()
    capi.batch_rows_scatter_add(T, 64, 5000, 5100, idx, idx, idx, 16, 50, DB.from_numpy(np.ones((48, 64), np.float32)))
    assert np.array_equal(T.numpy(), before)
