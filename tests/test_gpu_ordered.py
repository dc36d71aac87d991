"""csrc/ordered.hip: the ordered scatter-add behind the parity mode of the batch-gradient kernels.  Its contract is an ORDER --
every destination row receives its slots in ascending slot order, class by class, fp32 adds without contraction -- so the
checker is numpy's unbuffered ``np.add.at`` (sequential in index order) and the comparison is bit for bit."""
import numpy as np
import pytest

from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import LightGCNTrainer, joint_norm_adjacency, ordered_reductions

from helpers import check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def _bits_differ(a, b):
    return int(np.count_nonzero(np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)))


@pytest.mark.parametrize("ld", [32, 64, 128, 256])
@pytest.mark.parametrize("n_slots,n_rows,class_size", [(1, 5, 0), (4096, 37, 0), (6000, 900, 2000), (3 * 2048, 3500, 2048), (50000, 11, 0)])
def test_ordered_scatter_is_np_add_at_bit_for_bit(ld, n_slots, n_rows, class_size):
    rng = np.random.default_rng(n_slots * 7 + ld)
    src = ((rng.random((n_slots, ld), dtype=np.float32) - 0.5) * np.float32(10) ** rng.integers(-6, 3, (n_slots, 1))).astype(np.float32)   # wide dynamic range: the order shows
    rows = rng.integers(0, n_rows, n_slots).astype(np.int32)
    rows[rng.random(n_slots) < 0.05] = -1                                                # slots without a destination
    out0 = rng.random((n_rows, ld), dtype=np.float32)
    want = np.zeros((n_rows, ld), np.float32)
    live = rows >= 0
    if class_size:
        for c in range((n_slots + class_size - 1) // class_size):
            sl = slice(c * class_size, min((c + 1) * class_size, n_slots))
            part = np.zeros((n_rows, ld), np.float32)
            np.add.at(part, rows[sl][live[sl]], src[sl][live[sl]])
            want = (want + part).astype(np.float32)
    else:
        np.add.at(want, rows[live], src[live])
    want = (out0 + want).astype(np.float32)
    d_out = DB.from_numpy(out0)
    ws = capi.OrderedScatter()
    capi.scatter_add_rows_ordered(DB.from_numpy(src), DB.from_numpy(rows), n_slots, ld, d_out, ws, class_size=class_size)
    got = d_out.numpy()
    check("ordered scatter vs np.add.at: 32-bit words that differ", _bits_differ(got, want), 0, inclusive=True, ctx=(ld, n_slots, n_rows, class_size))
    # and again through the same (now warm) workspace: same bits
    d_out.upload(out0)
    capi.scatter_add_rows_ordered(DB.from_numpy(src), DB.from_numpy(rows), n_slots, ld, d_out, ws, class_size=class_size)
    check("ordered scatter, second launch: words that differ from the first", _bits_differ(d_out.numpy(), got), 0, inclusive=True)


def test_ordered_scatter_abi_errors():
    src, rows, out = DB.zeros((8, 64), np.float32), DB.zeros(8, np.int32), DB.zeros((4, 64), np.float32)
    small = DB.zeros(64, np.uint8)
    lib = capi.load()
    assert lib.qrec_scatter_add_rows_ordered(src.ptr, rows.ptr, 8, 64, 0, out.ptr, small.ptr, small.nbytes, None) < 0
    assert "workspace" in lib.qrec_last_error().decode()
    assert lib.qrec_scatter_add_rows_ordered(src.ptr, rows.ptr, 8, 48, 0, out.ptr, small.ptr, small.nbytes, None) < 0      # stride 48
    assert lib.qrec_scatter_add_rows_ordered(src.ptr, rows.ptr, 8, 64, 0, None, small.ptr, small.nbytes, None) < 0        # no destination
    assert lib.qrec_scatter_add_rows_ordered(None, None, 0, 64, 0, out.ptr, None, 0, None) == 0                           # nothing to do
    # the batch kernels refuse a workspace that is too small instead of writing past it
    idx = DB.zeros(16, np.int32); T = DB.zeros((100, 64), np.float32); loss = DB.zeros(1, np.float64)
    assert lib.qrec_bpr_batch_loss_grad(T.ptr, 1.0, 50, 100, 64, idx.ptr, idx.ptr, idx.ptr, 16, 1e-7, 0.0, T.ptr, loss.ptr, None, small.ptr, small.nbytes, None) < 0
    assert "workspace" in lib.qrec_last_error().decode()


def test_batch_gradient_in_parity_mode_equals_the_slot_order_sum_bit_for_bit():
    """qrec_bpr_batch_loss_grad with an ordered workspace: the table gradient is (sum of the u-lookups' rows) for user rows and
    (sum of the i-lookups' rows) + (sum of the j-lookups' rows) for item rows, each in batch order -- restated here in numpy from
    the same fp32 formulas (contraction off), compared bit for bit; the atomic mode on the same batch agrees to rounding only."""
    rng = np.random.default_rng(5)
    nu, ni, d, B = 300, 200, 64, 4096                                   # every row is looked up many times
    S = (rng.random((nu + ni, d), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
    u = rng.integers(0, nu, B).astype(np.int32); i = rng.integers(0, ni, B).astype(np.int32); j = rng.integers(0, ni, B).astype(np.int32)
    reg, eps = np.float32(0.01), np.float32(1e-7)
    d_S, d_loss = DB.from_numpy(S), DB.zeros(1, np.float64)
    outs = []
    for ordered in (capi.OrderedScatter(), capi.OrderedScatter(), None):
        dE = DB.zeros((nu + ni, d), np.float32)
        capi.bpr_batch_loss_grad(d_S, 1.0, nu, nu + ni, d, DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B, float(eps), float(reg), dE, d_loss, ordered=ordered)
        outs.append(dE.numpy())
    check("parity-mode batch gradient, two launches: words that differ", _bits_differ(outs[0], outs[1]), 0, inclusive=True)
    # the slots' rows from the device's own coefficient would need a read-back; instead the sums are re-formed from the device result's
    # definition with the coefficient recomputed in fp64 and the comparison made at fp32 rounding of ONE row sum -- the bitwise
    # statement is the two-launch one above; this one pins WHAT is summed
    ub, ib, jb = S[u].astype(np.float64), S[nu + i].astype(np.float64), S[nu + j].astype(np.float64)
    x = (ub * ib).sum(1) - (ub * jb).sum(1)
    sg = 1.0 / (1.0 + np.exp(-x))
    g = -(sg * (1.0 - sg)) / (sg + float(eps))
    want = np.zeros((nu + ni, d), np.float64)
    np.add.at(want, u, g[:, None] * (ib - jb) + float(reg) * ub)
    np.add.at(want, nu + i, g[:, None] * ub + float(reg) * ib)
    np.add.at(want, nu + j, -g[:, None] * ub + float(reg) * jb)
    for name, got in (("parity mode", outs[0]), ("atomic mode", outs[2])):
        check(f"batch gradient, {name}, vs the fp64 sum", float(np.linalg.norm(got - want) / np.linalg.norm(want)), 1e-6)


def test_a_trainer_built_under_ordered_reductions_is_bit_reproducible_and_one_built_outside_keeps_the_atomics():
    rng = np.random.default_rng(11)
    nu, ni, d, E, B = 400, 300, 32, 6000, 2048
    uid = rng.integers(0, nu, E); iid = rng.integers(0, ni, E)
    adj = joint_norm_adjacency(nu, ni, uid, iid)
    U0 = (rng.standard_normal((nu, d)) * 0.1).astype(np.float32); V0 = (rng.standard_normal((ni, d)) * 0.1).astype(np.float32)
    u = rng.integers(0, nu, B).astype(np.int32); i = rng.integers(0, ni, B).astype(np.int32); j = rng.integers(0, ni, B).astype(np.int32)

    def run():
        tr = LightGCNTrainer(U0, V0, adj, 2, lr=0.01, reg=1e-4)
        for _ in range(5):
            tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B)
        return tr, np.concatenate(tr.ego_embeddings())
    with ordered_reductions():
        (ta, a), (tb, b) = run(), run()
    assert ta.ows is not None and tb.ows is not None
    check("LightGCN, 5 steps, parity mode, two runs: words that differ", _bits_differ(a, b), 0, inclusive=True)
    tc, c = run()
    assert tc.ows is None                                              # outside the context: throughput mode, float atomics
    check("LightGCN, 5 steps: atomic mode vs parity mode", float(np.linalg.norm(c - a) / np.linalg.norm(a)), 1e-5)
