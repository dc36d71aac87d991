"""Checks of the numpy restatement of the TF-path models (oracle/tfmodels.py):
adjacency pinned to the live reference's scipy output; hand-derived gradients vs torch-CPU
autograd of the same graph; Adam vs torch.optim.Adam's closed form where they coincide."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import tfmodels as T

from helpers import load_golden


def test_adjacency_matches_reference_scipy_output():
    meta, z = load_golden("pairwise_adj_filmtrust")
    A = T.joint_norm_adjacency(meta["n_users"], meta["n_items"], z["train_uid"], z["train_iid"])
    A.sort_indices()
    assert A.dtype == np.float32 and A.nnz == meta["adj_nnz"]
    assert np.array_equal(A.indptr, z["adj_indptr"]) and np.array_equal(A.indices, z["adj_indices"])
    assert np.array_equal(A.data, z["adj_data"])            # bit-exact fp32 values
    # duplicated (u,i) rows in the file sum to 2 before normalisation (csr_matrix semantics)
    assert (A != A.T).nnz == 0


def _torch_lightgcn_loss(E, adj_t, nu, L, u, i, j, reg):
    layers = [E]
    for _ in range(L):
        layers.append(torch.sparse.mm(adj_t, layers[-1]))
    Ebar = torch.stack(layers).mean(0)
    ub, ib, jb = Ebar[u], Ebar[i + nu], Ebar[j + nu]
    score = (ub * ib).sum(1) - (ub * jb).sum(1)
    loss = -torch.log(torch.sigmoid(score) + 1e-7).sum()
    return loss + reg * 0.5 * ((ub ** 2).sum() + (ib ** 2).sum() + (jb ** 2).sum())


@pytest.mark.parametrize("L", [1, 2, 3])
def test_lightgcn_gradient_matches_autograd(L):
    rng = np.random.default_rng(L)
    nu, ni, d, B = 40, 30, 8, 64
    uid = rng.integers(0, nu, 300); iid = rng.integers(0, ni, 300)
    adj = T.joint_norm_adjacency(nu, ni, uid, iid)
    U0 = rng.standard_normal((nu, d)).astype(np.float32) * 0.1; V0 = rng.standard_normal((ni, d)).astype(np.float32) * 0.1
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B); j = rng.integers(0, ni, B)   # duplicates on purpose
    m = T.LightGCN(U0, V0, adj, L, lr=0.001, reg=0.01)
    loss, g = m.loss_and_grad(u, i, j)
    coo = adj.tocoo()
    adj_t = torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), adj.shape).coalesce()
    E = torch.tensor(np.concatenate([U0, V0]).astype(np.float64), requires_grad=True)
    tl = _torch_lightgcn_loss(E, adj_t, nu, L, torch.tensor(u), torch.tensor(i), torch.tensor(j), 0.01)
    tl.backward()
    assert loss == pytest.approx(float(tl), rel=2e-6)
    np.testing.assert_allclose(g, E.grad.numpy(), rtol=2e-4, atol=2e-6)


def test_adam_matches_closed_form():
    """fp32 ApplyAdam form vs the textbook formula in float64 (they differ only by rounding)."""
    rng = np.random.default_rng(0)
    theta = rng.standard_normal((5, 4)).astype(np.float32); th = theta.astype(np.float64)
    opt = T.AdamTF114(theta.shape, lr=0.01)
    m = np.zeros_like(th); v = np.zeros_like(th)
    for t in range(1, 8):
        g = rng.standard_normal(theta.shape).astype(np.float32)
        opt.step(theta, g)
        g = g.astype(np.float64)
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        th -= 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(theta, th, rtol=2e-4, atol=1e-6)


def test_lightgcn_training_reduces_loss_and_bpr_grad_finite_difference():
    rng = np.random.default_rng(3)
    nu, ni, d = 50, 40, 8
    uid = rng.integers(0, nu, 400); iid = rng.integers(0, ni, 400)
    adj = T.joint_norm_adjacency(nu, ni, uid, iid)
    m = T.LightGCN(rng.standard_normal((nu, d)).astype(np.float32) * 0.01, rng.standard_normal((ni, d)).astype(np.float32) * 0.01, adj, 2, 0.01, 1e-4)
    j = rng.integers(0, ni, 400)
    first = m.train_step(uid, iid, j)
    for _ in range(30):
        last = m.train_step(uid, iid, j)
    assert last < first
    # finite differences on the batch loss (float64 re-evaluation)
    ub, ib, jb = [rng.standard_normal((6, d)) for _ in range(3)]
    loss, du, di, dj = T.bpr_batch_loss_and_grads(ub.astype(np.float32), ib.astype(np.float32), jb.astype(np.float32), 0.05)
    def f(ub, ib, jb):
        s = 1 / (1 + np.exp(-((ub * ib).sum(1) - (ub * jb).sum(1))))
        return -np.log(s + 1e-7).sum() + 0.05 * 0.5 * ((ub ** 2).sum() + (ib ** 2).sum() + (jb ** 2).sum())
    h = 1e-6
    for arr, grad in ((ub, du), (ib, di), (jb, dj)):
        for idx in [(0, 0), (3, 5), (5, 7)]:
            a = arr.copy(); a[idx] += h
            args = [a if x is arr else x for x in (ub, ib, jb)]
            num = (f(*args) - f(ub, ib, jb)) / h
            assert grad[idx] == pytest.approx(num, rel=2e-3, abs=2e-5)


def _torch_simgcl_loss(E, adj_t, nu, L, u, i, j, reg, cl_rate, eps, noises):
    def enc(ns):
        emb, outs = E, []
        for k in range(L):
            emb = torch.sparse.mm(adj_t, emb)
            if ns is not None:
                nz = torch.nn.functional.normalize(ns[k], dim=1, eps=1e-6)
                emb = emb + torch.sign(emb) * nz * eps
            outs.append(emb)
        return torch.stack(outs).mean(0)
    main, p1, p2 = enc(None), enc(noises[:L]), enc(noises[L:])
    ub, ib, jb = main[u], main[i + nu], main[j + nu]
    score = (ub * ib).sum(1) - (ub * jb).sum(1)
    rec = -torch.log(torch.sigmoid(score) + 1e-7).sum() + reg * 0.5 * ((ub ** 2).sum() + (ib ** 2).sum() + (jb ** 2).sum())
    cl = 0
    for rows in (torch.unique(u), torch.unique(i) + nu):
        z1 = torch.nn.functional.normalize(p1[rows], dim=1, eps=1e-6); z2 = torch.nn.functional.normalize(p2[rows], dim=1, eps=1e-6)
        pos = torch.exp((z1 * z2).sum(1) / 0.2); ttl = torch.exp(z1 @ z2.T / 0.2).sum(1)
        cl = cl - torch.log(pos / ttl).sum()
    return rec + cl_rate * cl, rec, cl_rate * cl


@pytest.mark.parametrize("L", [1, 2, 3])
def test_simgcl_gradient_matches_autograd(L):
    rng = np.random.default_rng(10 + L)
    nu, ni, d, B = 40, 30, 8, 48
    uid = rng.integers(0, nu, 300); iid = rng.integers(0, ni, 300)
    adj = T.joint_norm_adjacency(nu, ni, uid, iid)
    U0 = rng.standard_normal((nu, d)).astype(np.float32) * 0.1; V0 = rng.standard_normal((ni, d)).astype(np.float32) * 0.1
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B); j = rng.integers(0, ni, B)
    noises = [rng.random((nu + ni, d)).astype(np.float32) for _ in range(2 * L)]
    m = T.SimGCL(U0, V0, adj, L, lr=0.001, reg=1e-3, cl_rate=0.5, eps=0.1)
    loss, rec, cl, g = m.loss_and_grad(u, i, j, noises)
    coo = adj.tocoo()
    adj_t = torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), adj.shape).coalesce()
    E = torch.tensor(np.concatenate([U0, V0]).astype(np.float64), requires_grad=True)
    tl, trec, tcl = _torch_simgcl_loss(E, adj_t, nu, L, torch.tensor(u), torch.tensor(i), torch.tensor(j), 1e-3, 0.5, 0.1,
                                       [torch.tensor(n.astype(np.float64)) for n in noises])
    tl.backward()
    assert rec == pytest.approx(float(trec.detach()), rel=1e-5) and cl == pytest.approx(float(tcl.detach()), rel=1e-5)
    np.testing.assert_allclose(g, E.grad.numpy(), rtol=2e-3, atol=2e-5)
    assert np.array_equal(T.unique_first_appearance([3, 1, 3, 2, 1]), [3, 1, 2])


def test_ngcf_gradients_match_autograd():
    rng = np.random.default_rng(42)
    nu, ni, d, B = 30, 25, 8, 40
    uid = rng.integers(0, nu, 200); iid = rng.integers(0, ni, 200)
    adj = T.joint_norm_adjacency(nu, ni, uid, iid)
    U0 = rng.standard_normal((nu, d)).astype(np.float32) * 0.3; V0 = rng.standard_normal((ni, d)).astype(np.float32) * 0.3
    W = [[rng.standard_normal((d, d)).astype(np.float32) * 0.4 for _ in range(2)] for _ in range(2)]
    masks = [(rng.random((nu + ni, d)) < 0.9).astype(np.float32) for _ in range(2)]
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B); j = rng.integers(0, ni, B)
    m = T.NGCF(U0, V0, W, adj, lr=0.002, reg=0.01)
    loss, gE, gW = m.loss_and_grads(u, i, j, masks)
    coo = adj.tocoo()
    A = torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), adj.shape).coalesce()
    E = torch.tensor(np.concatenate([U0, V0]).astype(np.float64), requires_grad=True)
    Wt = [[torch.tensor(w.astype(np.float64), requires_grad=True) for w in pair] for pair in W]
    cur, outs = E, [E]
    for k in range(2):
        side = torch.sparse.mm(A, cur)
        pre = (side + cur) @ Wt[k][0] + (cur * side) @ Wt[k][1]
        cur = torch.nn.functional.leaky_relu(pre, 0.2) * torch.tensor(masks[k].astype(np.float64)) / 0.9
        outs.append(torch.nn.functional.normalize(cur, dim=1, eps=1e-6))
    allE = torch.cat(outs, 1)
    ub, ib, jb = allE[torch.tensor(u)], allE[torch.tensor(i) + nu], allE[torch.tensor(j) + nu]
    score = (ub * ib).sum(1) - (ub * jb).sum(1)
    tl = -torch.log(torch.sigmoid(score) + 1e-7).sum() + 0.01 * 0.5 * ((ub ** 2).sum() + (ib ** 2).sum() + (jb ** 2).sum())
    tl.backward()
    assert loss == pytest.approx(float(tl.detach()), rel=1e-5)
    np.testing.assert_allclose(gE, E.grad.numpy(), rtol=2e-3, atol=2e-5)
    for k in range(2):
        for t in range(2):
            np.testing.assert_allclose(gW[k][t], Wt[k][t].grad.numpy(), rtol=2e-3, atol=2e-5)
    first = m.train_step(u, i, j, masks)
    for _ in range(20):
        last = m.train_step(u, i, j, masks)
    assert last < first
    Uf, Vf = m.inference_embeddings(); assert Uf.shape == (nu, 3 * d) and Vf.shape == (ni, 3 * d)


@pytest.mark.parametrize("per_layer", [False, True])
def test_sgl_gradient_matches_autograd(per_layer):
    rng = np.random.default_rng(77)
    nu, ni, d, B, L = 40, 30, 8, 48, 2
    uid = rng.integers(0, nu, 400); iid = rng.integers(0, ni, 400)
    adj = T.joint_norm_adjacency(nu, ni, uid, iid)
    def sub():
        keep = rng.permutation(400)[:360]
        return T.joint_norm_adjacency(nu, ni, uid[keep], iid[keep])
    mats1 = [sub() for _ in range(L)] if per_layer else [sub()] * L
    mats2 = [sub() for _ in range(L)] if per_layer else [sub()] * L
    U0 = rng.standard_normal((nu, d)).astype(np.float32) * 0.1; V0 = rng.standard_normal((ni, d)).astype(np.float32) * 0.1
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B); j = rng.integers(0, ni, B)
    m = T.SGL(U0, V0, adj, L, lr=0.001, reg=1e-3, ssl_reg=0.1, temp=0.2)
    loss, rec, ssl, g = m.loss_and_grad(u, i, j, mats1, mats2)
    def tsp(a):
        coo = a.tocoo(); return torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), a.shape).coalesce()
    E = torch.tensor(np.concatenate([U0, V0]).astype(np.float64), requires_grad=True)
    def view(mats):
        layers = [E]
        for k in range(L):
            layers.append(torch.sparse.mm(tsp(mats[k]), layers[-1]))
        return torch.stack(layers).mean(0)
    main, s1, s2 = view([adj] * L), view(mats1), view(mats2)
    tu, ti, tj = torch.tensor(u), torch.tensor(i), torch.tensor(j)
    ub, ib, jb = main[tu], main[ti + nu], main[tj + nu]
    trec = -torch.log(torch.sigmoid((ub * ib).sum(1) - (ub * jb).sum(1)) + 1e-7).sum() + 1e-3 * 0.5 * ((ub ** 2).sum() + (ib ** 2).sum() + (jb ** 2).sum())
    rows = torch.cat([torch.unique(tu), torch.unique(ti) + nu])
    z1 = torch.nn.functional.normalize(s1[rows], dim=1, eps=1e-6); z2 = torch.nn.functional.normalize(s2[rows], dim=1, eps=1e-6)
    tssl = -0.1 * torch.log(torch.exp((z1 * z2).sum(1) / 0.2) / torch.exp(z1 @ z2.T / 0.2).sum(1)).sum()
    (trec + tssl).backward()
    assert rec == pytest.approx(float(trec.detach()), rel=1e-5) and ssl == pytest.approx(float(tssl.detach()), rel=1e-5)
    np.testing.assert_allclose(g, E.grad.numpy(), rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("L", [1, 2])
def test_buir_gradient_matches_autograd(L):
    """model/ranking/BUIR.py:88-130: hand-derived gradients of the online tables, W and b vs torch autograd; the
    target side carries no gradient; the momentum update follows the Adam step."""
    rng = np.random.default_rng(91)
    nu, ni, d, B = 40, 30, 8, 48
    uid = rng.integers(0, nu, 400); iid = rng.integers(0, ni, 400)
    adj = T.joint_norm_adjacency(nu, ni, uid, iid)
    def sub():
        keep = rng.permutation(400)[:200]
        return T.joint_norm_adjacency(nu, ni, uid[keep], iid[keep])
    mo, mt = sub(), sub()
    U0 = rng.standard_normal((nu, d)).astype(np.float32) * 0.3; V0 = rng.standard_normal((ni, d)).astype(np.float32) * 0.3
    W0 = rng.standard_normal((d, d)).astype(np.float32) * 0.4; b0 = rng.standard_normal((1, d)).astype(np.float32) * 0.1
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B)
    m = T.BUIR(U0, V0, W0, b0, L, lr=0.001, tau=0.995)
    m.T = (m.T + rng.standard_normal(m.T.shape).astype(np.float32) * 0.05).astype(np.float32)   # target != online, as after some steps
    loss, gE, gW, gb = m.loss_and_grads(u, i, mo, mt)
    def tsp(a):
        coo = a.tocoo(); return torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), a.shape).coalesce()
    E = torch.tensor(m.E.astype(np.float64), requires_grad=True); Tt = torch.tensor(m.T.astype(np.float64))
    W = torch.tensor(W0.astype(np.float64), requires_grad=True); b = torch.tensor(b0.astype(np.float64), requires_grad=True)
    def mean_prop(mat, X):
        layers = [X]
        for _ in range(L):
            layers.append(torch.sparse.mm(tsp(mat), layers[-1]))
        return torch.stack(layers).mean(0)
    online, target = mean_prop(mo, E), mean_prop(mt, Tt)
    q = torch.tanh(online @ W + b)
    tu, ti = torch.tensor(u), torch.tensor(i) + nu
    nz = lambda x: torch.nn.functional.normalize(x, dim=1, eps=1e-6)
    tl = ((1 - (nz(q[tu]) * nz(target[ti])).sum(1)) + (1 - (nz(q[ti]) * nz(target[tu])).sum(1))).sum() / 2
    tl.backward()
    assert loss == pytest.approx(float(tl.detach()), rel=1e-5)
    np.testing.assert_allclose(gE, E.grad.numpy(), rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(gW, W.grad.numpy(), rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(gb, b.grad.numpy(), rtol=2e-3, atol=2e-6)
    E_before, T_before = m.E.copy(), m.T.copy()
    first = m.train_step(u, i, mo, mt)
    np.testing.assert_allclose(m.T, T_before * np.float32(0.995) + m.E * np.float32(0.005), rtol=1e-6)   # EMA of the UPDATED online tables
    assert not np.array_equal(m.E, E_before)
    for _ in range(30):
        last = m.train_step(u, i, mo, mt)
    assert last < first
    qu, qi, ou, oi = m.final_tables(adj)
    assert qu.shape == (nu, d) and qi.shape == (ni, d) and ou.shape == (nu, d) and oi.shape == (ni, d)


@pytest.mark.parametrize("joint", [False, True])
def test_sept_gradient_matches_autograd(joint):
    """model/ranking/SEPT.py:124-270: hand-derived gradient of the two variables through the four normalised-layer
    views, the several-positives contrastive loss and the Variable/2 scaling vs torch autograd of the same graph (the
    pseudo-label indices are data: taken from the restatement's own top-k); the two Adam optimizers keep separate slots."""
    rng = np.random.default_rng(131)
    nu, ni, d, B, L, k = 40, 50, 8, 64, 2, 5
    uid = rng.integers(0, nu, 300); iid = rng.integers(0, ni, 300)
    fo = rng.integers(0, nu, 200); fe = rng.integers(0, nu, 200)
    social, sharing = T.sept_social_views(nu, ni, uid, iid, fo, fe)
    adj = T.sept_sub_adjacency(nu, ni, uid, iid, fo, fe)
    sub = T.sept_sub_adjacency(nu, ni, uid, iid, fo, fe, rng.permutation(300)[:210], rng.permutation(200)[:140])
    assert abs(sub - sub.T).max() > 0                              # the follow edges make it non-symmetric: backward uses M^T
    U0 = (rng.standard_normal((nu, d)) * 0.2).astype(np.float32); V0 = (rng.standard_normal((ni, d)) * 0.2).astype(np.float32)
    m = T.SEPT(U0, V0, adj, social, sharing, L, lr=0.001, reg=0.01, ss_rate=0.05, ins_cnt=k)
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B); j = rng.integers(0, ni, B)
    rec, nd, g, labels = m.loss_and_grad(u, i, j, sub if joint else None)

    def tsp(a):
        coo = a.tocoo(); return torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), a.shape).coalesce()
    nz = lambda x: torch.nn.functional.normalize(x, dim=1, eps=1e-6)
    W = torch.tensor(m.W.astype(np.float64), requires_grad=True)
    E0 = W / 2

    def view(mat, X):
        out, cur = [X], X
        for _ in range(L):
            cur = torch.sparse.mm(tsp(mat), cur); out.append(nz(cur))
        return torch.stack(out).sum(0)
    Se = view(m.adj, E0)
    tu, ti, tj = torch.tensor(u), torch.tensor(i) + nu, torch.tensor(j) + nu
    trec = -torch.log(torch.sigmoid((Se[tu] * Se[ti]).sum(1) - (Se[tu] * Se[tj]).sum(1)) + 1e-7).sum() + 0.01 * 0.5 * (E0 ** 2).sum()
    tnd = torch.zeros((), dtype=torch.float64)
    if joint:
        rows = torch.tensor(T.unique_first_appearance(u))
        a = nz(view(sub.astype(np.float32), E0)[rows])
        for S, pos in zip((view(m.social, E0[:nu]), view(m.sharing, E0[:nu]), Se), labels):
            z = nz(S[rows]); e = torch.exp(z @ a.T / 0.1)
            tnd = tnd - torch.log(torch.gather(e, 1, torch.tensor(pos)).sum(1) / e.sum(1)).sum()
        assert all(p.shape == (rows.numel(), k) and (np.sort(p, 1)[:, 1:] != np.sort(p, 1)[:, :-1]).all() for p in labels)
    (trec + 0.05 * tnd).backward()
    assert rec == pytest.approx(float(trec.detach()), rel=1e-5) and nd == pytest.approx(float(tnd.detach()), rel=1e-4, abs=1e-9)
    np.testing.assert_allclose(g, W.grad.numpy(), rtol=3e-3, atol=3e-6)
    # pseudo labels: the k best of the averaged softmax rows, ties in index order
    s = np.array([[0.2, 0.5, 0.5, 0.1, 0.5], [0.3, 0.3, 0.1, 0.9, 0.0]], np.float32)
    assert T.top_k_rows(s, 3).tolist() == [[1, 2, 4], [3, 0, 1]]
    # two optimizers, separate slots and step counts (v1_op before maxEpoch/3, v2_op after)
    first = m.train_step(u, i, j)
    assert m.opt1.t == 1 and m.opt2.t == 0 and not m.opt2.m.any()
    m.train_step(u, i, j, sub)
    assert m.opt1.t == 1 and m.opt2.t == 1 and m.opt2.m.any()
    for _ in range(25):
        last = m.train_step(u, i, j)
    assert last[0] < first[0] and last[1] == 0.0


def _mhcn_problem(rng, nu=40, ni=50, d=8, E=400, R=260):
    uid = rng.integers(0, nu, E); iid = rng.integers(0, ni, E)
    pairs = np.unique(np.stack([uid, iid], 1), axis=0); uid, iid = pairs[:, 0], pairs[:, 1]
    fo = rng.integers(0, nu, R); fe = rng.integers(0, nu, R)
    keep = fo != fe
    rel = np.unique(np.stack([fo[keep], fe[keep]], 1), axis=0)
    H = T.mhcn_motif_adjacencies(nu, ni, uid, iid, rel[:, 0], rel[:, 1])
    Rm = T.mhcn_joint_adjacency(nu, ni, uid, iid, np.ones(uid.size))
    lim = np.sqrt(6.0 / (2 * d))
    w = {}
    for k in (1, 2, 3, 4):
        for pre in ("gating", "sgating"):
            w[f"{pre}{k}"] = rng.uniform(-lim, lim, (d, d)).astype(np.float32)
            w[f"{pre}_bias{k}"] = rng.uniform(-0.5, 0.5, (1, d)).astype(np.float32)
    w["attention"] = rng.uniform(-0.5, 0.5, (1, d)).astype(np.float32); w["attention_mat"] = rng.uniform(-lim, lim, (d, d)).astype(np.float32)
    U0 = (rng.standard_normal((nu, d)) * 0.3).astype(np.float32); V0 = (rng.standard_normal((ni, d)) * 0.3).astype(np.float32)
    perms = [(rng.permutation(nu), rng.permutation(d), rng.permutation(nu), rng.permutation(d), rng.permutation(nu)) for _ in range(3)]
    return nu, ni, d, H, Rm, w, U0, V0, perms


@pytest.mark.parametrize("L", [1, 2])
def test_mhcn_gradients_match_autograd(L):
    """model/ranking/MHCN.py:93-216: hand-derived gradients of U, V and all 18 weight tensors (self-gating, channel
    attention, three hypergraph channels + the user-item channel with per-layer normalisation, hierarchical mutual
    information with injected shuffles, weight and table L2) vs torch autograd of the same graph."""
    rng = np.random.default_rng(170 + L)
    nu, ni, d, H, Rm, w, U0, V0, perms = _mhcn_problem(rng)
    assert all(h.nnz > 0 for h in H)
    m = T.MHCN(U0, V0, w, H, Rm, L, lr=0.001, reg=0.01, ss_rate=0.05)
    B = 64
    u = rng.integers(0, nu, B); i = rng.integers(0, ni, B); j = rng.integers(0, ni, B)
    rec, ss, reg, g = m.loss_and_grads(u, i, j, perms)

    def tsp(a):
        coo = a.tocoo(); return torch.sparse_coo_tensor(np.vstack([coo.row, coo.col]), coo.data.astype(np.float64), a.shape).coalesce()
    tw = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in w.items()}
    tU = torch.tensor(U0.astype(np.float64), requires_grad=True); tV = torch.tensor(V0.astype(np.float64), requires_grad=True)
    tH = [tsp(h) for h in m.H]; tR = tsp(m.R); tRT = tsp(m.R.T.tocsr())
    nz = lambda x: torch.nn.functional.normalize(x, dim=1, eps=1e-6)
    gate = lambda x, pre, k: x * torch.sigmoid(x @ tw[f"{pre}{k}"] + tw[f"{pre}_bias{k}"])

    def att(es):
        ws = torch.stack([(tw["attention"] * (e @ tw["attention_mat"])).sum(1) for e in es], 1)
        sc = torch.softmax(ws, 1)
        return sum(sc[:, k:k + 1] * es[k] for k in range(3))
    c = [gate(tU, "gating", k) for k in (1, 2, 3)]; s = gate(tU, "gating", 4); t = tV
    allc = [[x] for x in c]; alls = [s]; allt = [t]
    for _ in range(L):
        mixed = att(c) + s / 2
        c = [torch.sparse.mm(tH[k], c[k]) for k in range(3)]
        for k in range(3):
            allc[k].append(nz(c[k]))
        t_new = torch.sparse.mm(tRT, mixed); allt.append(nz(t_new))
        s = torch.sparse.mm(tR, t); alls.append(nz(s))
        t = t_new
    fu = att([torch.stack(x).sum(0) for x in allc]) + torch.stack(alls).sum(0) / 2
    fi = torch.stack(allt).sum(0)
    tss = torch.zeros((), dtype=torch.float64)
    ls = lambda x: -torch.log(torch.sigmoid(x))
    for k in range(3):
        em = gate(fu, "sgating", k + 1)
        p1, k2, p2, k3, p3 = (torch.tensor(x) for x in perms[k])
        edge = torch.sparse.mm(tH[k], em)
        pos, neg1, neg2 = (em * edge).sum(1), (em[p1] * edge).sum(1), (edge[:, k2][p2] * em).sum(1)
        graph = edge.mean(0)
        tss = tss + (ls(pos - neg1) + ls(neg1 - neg2)).sum() + ls(edge @ graph - edge[:, k3][p3] @ graph).sum()
    tu, ti, tj = torch.tensor(u), torch.tensor(i), torch.tensor(j)
    trec = -torch.log(torch.sigmoid((fu[tu] * fi[ti]).sum(1) - (fu[tu] * fi[tj]).sum(1)) + 1e-7).sum()
    treg = 0.001 * sum(0.5 * (v ** 2).sum() for v in tw.values()) + 0.01 * 0.5 * ((tU ** 2).sum() + (tV ** 2).sum())
    (trec + treg + 0.05 * tss).backward()
    assert rec == pytest.approx(float(trec.detach()), rel=1e-5) and ss == pytest.approx(float(tss.detach()), rel=1e-5)
    assert reg == pytest.approx(float(treg.detach()), rel=1e-6)
    for k, v in tw.items():
        np.testing.assert_allclose(g[k], v.grad.numpy(), rtol=3e-3, atol=3e-6, err_msg=k)
    np.testing.assert_allclose(g["U"], tU.grad.numpy(), rtol=3e-3, atol=3e-6)
    np.testing.assert_allclose(g["V"], tV.grad.numpy(), rtol=3e-3, atol=3e-6)
    # sgating4 exists (n_channel = 4) but only three self-supervised gates are used: it sees its L2 term alone
    np.testing.assert_allclose(g["sgating4"], np.float32(0.001) * w["sgating4"], rtol=1e-6)
    first = m.train_step(u, i, j, perms)
    for _ in range(30):
        last = m.train_step(u, i, j, perms)
    assert last < first
