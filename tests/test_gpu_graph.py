"""GPU parity tests of the LightGCN-family kernels against the numpy/scipy restatement
(oracle/tfmodels.py): SpMM, batch BPR loss/grad, Adam, whole training steps, the drop-in class.
fp32 tolerance 1e-5 relative (north_star)."""
import io
import random
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tfmodels as T
from qrec_amd import capi
from qrec_amd.capi import DeviceBuffer as DB
from qrec_amd.graph import LightGCNTrainer, SpmmPlan, joint_norm_adjacency
from qrec_amd.synth import make_dataset

from helpers import conf_from_text, load_golden, pad_cols, rel_err, rows_from_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def _graph(shape):
    d = make_dataset(shape)
    adj = joint_norm_adjacency(d["n_users"], d["n_items"], d["train_u"], d["train_i"])
    A = sp.csr_matrix((adj[2], adj[1], adj[0]), shape=(adj[0].size - 1,) * 2)
    return d, adj, A


@pytest.mark.parametrize("dim,ld", [(64, 64), (50, 64), (8, 32), (128, 128), (200, 256)])
@pytest.mark.parametrize("seg_len", [128, 7])
def test_spmm_matches_scipy(dim, ld, seg_len):
    d, adj, A = _graph("small")
    n = A.shape[0]
    rng = np.random.default_rng(dim)
    X = rng.standard_normal((n, dim)).astype(np.float32); Z = rng.standard_normal((n, dim)).astype(np.float32)
    S0 = rng.standard_normal((n, dim)).astype(np.float32)
    plan = SpmmPlan(adj[0], adj[1], adj[2], ld, seg_len=seg_len)
    assert plan.n_long > 0 if seg_len == 7 else True
    dX, dY = DB.from_numpy(pad_cols(X, ld)), DB.zeros((n, ld), np.float32)
    capi.spmm_csr(plan, dX, dY, ld)
    ref = A.dot(X)
    got = dY.numpy()
    assert rel_err(got[:, :dim], ref) < TOL and (got[:, dim:] == 0).all()
    # rows that are not segmented accumulate in CSR order exactly like scipy: bit-identical
    whole = np.diff(adj[0]) <= seg_len
    assert np.array_equal(got[whole][:, :dim], ref[whole])
    # fused epilogues: Y = A X + 0.5 Z ; accum += Y
    dZ, dS = DB.from_numpy(pad_cols(Z, ld)), DB.from_numpy(pad_cols(S0, ld))
    capi.spmm_csr(plan, dX, dY, ld, d_addend=dZ, addend_scale=0.5, d_accum=dS)
    assert rel_err(dY.numpy()[:, :dim], ref + np.float32(0.5) * Z) < TOL
    assert rel_err(dS.numpy()[:, :dim], S0 + (ref + np.float32(0.5) * Z)) < TOL
    # deterministic
    capi.spmm_csr(plan, dX, dY, ld); a = dY.numpy(); capi.spmm_csr(plan, dX, dY, ld); assert np.array_equal(a, dY.numpy())
    with pytest.raises(capi.QRecError):
        capi.spmm_csr(plan, dX, dX, ld)


def test_spmm_empty_rows_and_heavy_row():
    # node 0 connected to everything (one very long row), nodes without edges (empty rows)
    n = 3000
    rows = np.concatenate([np.zeros(n - 10, np.int64), np.arange(10, n, dtype=np.int64)])
    cols = np.concatenate([np.arange(10, n, dtype=np.int64), np.zeros(n - 10, np.int64)])
    A = sp.csr_matrix((np.full(rows.size, 0.5, np.float32), (rows, cols)), shape=(n, n)); A.sort_indices()
    plan = SpmmPlan(A.indptr.astype(np.int64), A.indices, A.data, 64)
    X = np.random.default_rng(0).standard_normal((n, 64)).astype(np.float32)
    dX, dY = DB.from_numpy(X), DB.from_numpy(np.full((n, 64), 7, np.float32))
    capi.spmm_csr(plan, dX, dY, 64)
    got = dY.numpy()
    assert rel_err(got, A.dot(X)) < TOL and (got[1:10] == 0).all()


def test_batch_loss_grad_and_adam_match_restatement():
    rng = np.random.default_rng(1)
    nu, ni, dim, ld, B, L = 300, 200, 50, 64, 1000, 2
    S = rng.standard_normal((nu + ni, dim)).astype(np.float32)
    u = rng.integers(0, nu, B).astype(np.int32); i = rng.integers(0, ni, B).astype(np.int32); j = rng.integers(0, ni, B).astype(np.int32)
    Ebar = (S / np.float32(L + 1)).astype(np.float32)
    loss, du, di, dj = T.bpr_batch_loss_and_grads(Ebar[u], Ebar[nu + i], Ebar[nu + j], 0.01)
    dref = np.zeros_like(Ebar); np.add.at(dref, u, du); np.add.at(dref, nu + i, di); np.add.at(dref, nu + j, dj)
    dS, dE, dl = DB.from_numpy(pad_cols(S, ld)), DB.zeros((nu + ni, ld), np.float32), DB.zeros(1, np.float64)
    capi.bpr_batch_loss_grad(dS, float(L + 1), nu, nu + ni, ld, DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B, 1e-7, 0.01, dE, dl)
    got = dE.numpy()
    assert rel_err(got[:, :dim], dref) < TOL and (got[:, dim:] == 0).all()
    assert abs(dl.numpy()[0] - loss) / abs(loss) < TOL
    # Adam, several steps
    theta = rng.standard_normal((nu + ni, ld)).astype(np.float32); ref = theta.copy()
    opt = T.AdamTF114(theta.shape, lr=0.01)
    dT, dM, dV = DB.from_numpy(theta), DB.zeros(theta.shape, np.float32), DB.zeros(theta.shape, np.float32)
    for t in range(1, 6):
        g = rng.standard_normal(theta.shape).astype(np.float32)
        alpha = float(opt.alpha())
        opt.step(ref, (np.float32(1 / 3) * g).astype(np.float32))
        capi.adam_step(dT, dM, dV, DB.from_numpy(g), theta.size, 1 / 3, alpha)
    assert rel_err(dT.numpy(), ref) < TOL and rel_err(dM.numpy(), opt.m) < TOL and rel_err(dV.numpy(), opt.v) < TOL


@pytest.mark.parametrize("L", [1, 2, 3])
def test_lightgcn_training_steps_match_restatement(L):
    d, adj, A = _graph("small")
    nu, ni, dim, B = d["n_users"], d["n_items"], 64, 2048
    rng = np.random.default_rng(L)
    U0 = (rng.standard_normal((nu, dim)) * 0.005).astype(np.float32); V0 = (rng.standard_normal((ni, dim)) * 0.005).astype(np.float32)
    ref = T.LightGCN(U0, V0, A, L, lr=0.001, reg=1e-4)
    tr = LightGCNTrainer(U0, V0, adj, L, lr=0.001, reg=1e-4)
    for step in range(8):
        sel = rng.integers(0, d["train_u"].size, B)
        u = d["train_u"][sel].astype(np.int32); i = d["train_i"][sel].astype(np.int32); j = rng.integers(0, ni, B).astype(np.int32)
        lref = ref.train_step(u, i, j)
        tr.train_step_async(DB.from_numpy(u), DB.from_numpy(i), DB.from_numpy(j), B)
        assert abs(tr.loss() - lref) / abs(lref) < TOL
    Ug, Vg = tr.ego_embeddings()
    # Adam normalises every coordinate's step to ~lr, so agreement is measured on the update
    assert rel_err(np.concatenate([Ug, Vg]) - np.concatenate([U0, V0]), ref.E - np.concatenate([U0, V0])) < 1e-3
    assert rel_err(np.concatenate([Ug, Vg]), ref.E) < TOL
    Uf, Vf = tr.final_embeddings(); Ur, Vr = ref.final_embeddings()
    assert rel_err(Uf, Ur) < 1e-4 and rel_err(Vf, Vr) < 1e-4


def test_lightgcn_class_end_to_end_against_restatement_with_reference_sampler_stream():
    """The drop-in LightGCN class on the reference's FilmTrust rows: its batches are the
    reference's own next_batch_pairwise stream (golden), the training trajectory equals the
    restatement's from the same injected initial tables."""
    from qrec_amd.model.ranking.LightGCN import LightGCN
    meta, z = load_golden("pairwise_adj_filmtrust")
    gz = load_golden("bpr_filmtrust")[1]
    train, test = rows_from_golden(gz)
    conf = conf_from_text(meta["conf"]); conf["num.max.epoch"] = "2"
    rng = np.random.default_rng(0)
    U0 = (rng.standard_normal((meta["n_users"], 8)) * 0.005).astype(np.float32)
    V0 = (rng.standard_normal((meta["n_items"], 8)) * 0.005).astype(np.float32)
    random.seed(meta["seed"]); np.random.seed(meta["seed"])
    buf = io.StringIO()
    with redirect_stdout(buf):
        m = LightGCN(conf, train, test)
        m.readConfiguration(); m.initializing_log = lambda: None
        m.initModel()
        m.trainer = type(m.trainer)(U0, V0, m.create_joint_sparse_adjaceny(), m.n_layers, m.lRate, m.regU)  # inject init
        m.trainModel()
    losses = [float(l.split("loss:")[1]) for l in buf.getvalue().splitlines() if "loss:" in l]
    # restatement fed with the reference's recorded stream (2 epochs were recorded)
    A = sp.csr_matrix((z["adj_data"], z["adj_indices"], z["adj_indptr"]), shape=(z["adj_indptr"].size - 1,) * 2)
    ref = T.LightGCN(U0, V0, A, 2, lr=float(conf["learnRate"].split()[1]), reg=1e-3)
    st, pos, ref_losses = z["stream"], 0, []
    for bs in z["batch_sizes"]:
        b = st[pos:pos + bs]; pos += bs
        ref_losses.append(ref.train_step(b[:, 0], b[:, 1], b[:, 2]))
    assert len(losses) == len(ref_losses)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    Ur, Vr = ref.final_embeddings()
    assert rel_err(m.U, Ur) < 1e-4 and rel_err(m.V, Vr) < 1e-4
    assert np.array_equal(capi.state_from_python(random.getstate()), z["py_state"])   # sampler stayed in lock-step
