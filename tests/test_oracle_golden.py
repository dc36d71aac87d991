"""Pin the CPU oracle (oracle/qrec_oracle.c) against golden vectors produced by the
UNMODIFIED reference run in-process (tests/golden/gen_golden.py).

Index streams must match bit-exact; fp64 state to 1e-11 relative (numpy's BLAS ddot sums
in a different order than the oracle's sequential loop, so bitwise equality of doubles is
not expected after thousands of dependent updates)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import c as O
from qrec_amd.interactions import user_item_csr


def _load(golden_dir, name):
    meta = json.load(open(os.path.join(golden_dir, "golden_meta.json")))[name]
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return meta, z


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _bpr_replay(meta, z, P, Q, mt, check_epoch):
    """Replay model/ranking/BPR.py:19-43 + base/iterativeRecommender.py:82-102 with the
    oracle; calls check_epoch(k, stream_k, P, Q, loss, lr_next) per epoch."""
    U, I = meta["n_users"], meta["n_items"]
    pos = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I, min_rating=1)
    u = pos.row_ids(); i = pos.indices
    assert pos.nnz == meta["triplets_per_epoch"]
    lr = meta["epochs"][0]["lr_used"]; max_lr = 1.0
    last_loss = 0.0
    streams = []
    for k, ep in enumerate(meta["epochs"]):
        j = O.bpr_sample_epoch(mt, pos.indptr, i, I)
        loss = O.bpr_sgd(P, Q, u, i, j, lr, meta["regU"], meta["regI"])
        loss += meta["regU"] * O.sumsq(P) + meta["regI"] * O.sumsq(Q)
        assert lr == pytest.approx(ep["lr_used"], rel=1e-15)
        # isConverged: lr update then shuffle(trainingData)
        delta = last_loss - loss
        if not abs(delta) < 1e-3:
            if ep["epoch"] > 1:
                lr = lr * 1.05 if abs(last_loss) > abs(loss) else lr * 0.5
            if lr > max_lr > 0:
                lr = max_lr
        last_loss = loss
        mt.shuffle(meta["n_train"])
        st = np.stack([u, i, j], axis=1)
        streams.append(st)
        check_epoch(k, st, P, Q, loss, lr, ep)
    return np.concatenate(streams)


def test_bpr_filmtrust_stream_and_state(golden_dir):
    meta, z = _load(golden_dir, "bpr_filmtrust")
    P, Q = z["P0"].copy(), z["Q0"].copy()
    # tables: np.random.seed(s); rand(U,d)/3; rand(I,d)/3 (base/iterativeRecommender.py:37-38)
    m_np = O.MT.numpy_seed(meta["seed"])
    assert np.array_equal(m_np.numpy_rand(*P.shape) / 3, P)
    assert np.array_equal(m_np.numpy_rand(*Q.shape) / 3, Q)
    mt = O.MT.cpython_seed(meta["seed"])

    def chk(k, st, P, Q, loss, lr, ep):
        np.testing.assert_allclose(P, z[f"P{k+1}"], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(Q, z[f"Q{k+1}"], rtol=1e-11, atol=1e-14)
        assert loss == pytest.approx(ep["loss"], rel=1e-12)
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)

    stream = _bpr_replay(meta, z, P, Q, mt, chk)
    assert np.array_equal(stream, z["steps"])                 # bit-exact index stream
    assert _sha(stream.astype(np.int32)) == meta["stream_sha256"]
    assert np.array_equal(mt.words625(), z["py_state"])       # generator ends in the same state


def test_bpr_lastfm_split_stream_and_state(golden_dir):
    meta, z = _load(golden_dir, "bpr_lastfm")
    seed = meta["seed"]
    mt = O.MT.cpython_seed(seed)
    # -ap 0.2 split: util/dataSplit.py:9-26, one random() per raw row
    mask = mt.data_split(z["split_is_test"].size, 0.2)
    assert np.array_equal(mask, z["split_is_test"])
    U, I, d = meta["n_users"], meta["n_items"], meta["emb_size"]
    m_np = O.MT.numpy_seed(seed)
    P = m_np.numpy_rand(U, d) / 3; Q = m_np.numpy_rand(I, d) / 3
    assert _sha(P) == meta["P0_sha256"] and _sha(Q) == meta["Q0_sha256"]
    last = len(meta["epochs"])

    def chk(k, st, P, Q, loss, lr, ep):
        assert loss == pytest.approx(ep["loss"], rel=1e-12)
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)
        if k + 1 == last:
            np.testing.assert_allclose(P, z[f"P{last}"], rtol=1e-11, atol=1e-14)
            np.testing.assert_allclose(Q[::4], z[f"Q{last}_every4"], rtol=1e-11, atol=1e-14)

    stream = _bpr_replay(meta, z, P, Q, mt, chk)
    assert _sha(stream.astype(np.int32)) == meta["stream_sha256"]
    assert np.array_equal(stream[:4096], z["steps_head"]) and np.array_equal(stream[-4096:], z["steps_tail"])
    assert np.array_equal(mt.words625(), z["py_state"])


def test_basicmf_filmtrust(golden_dir):
    """BASELINE.json config #1 through the oracle: model/rating/BasicMF.py:9-26."""
    meta, z = _load(golden_dir, "basicmf_filmtrust")
    P, Q = z["P0"].copy(), z["Q0"].copy()
    mt = O.MT.cpython_seed(meta["seed"])
    n = z["order0"].shape[0]
    perm = np.arange(n, dtype=np.int64)
    u0 = np.ascontiguousarray(z["order0"][:, 0]); i0 = np.ascontiguousarray(z["order0"][:, 1]); r0 = z["rating0"]
    lr = meta["epochs"][0]["lr_used"]; last_loss = 0.0
    for k, ep in enumerate(meta["epochs"]):
        u = np.ascontiguousarray(u0[perm]); i = np.ascontiguousarray(i0[perm]); r = np.ascontiguousarray(r0[perm])
        assert np.array_equal(np.stack([u, i], 1), z[f"order{k}"])
        loss = O.mf_sgd(P, Q, u, i, r, lr)
        assert loss == pytest.approx(ep["loss"], rel=1e-11)
        np.testing.assert_allclose(P, z[f"P{k+1}"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(Q, z[f"Q{k+1}"], rtol=1e-10, atol=1e-13)
        if not abs(last_loss - loss) < 1e-3:
            if ep["epoch"] > 1:
                lr = lr * 1.05 if abs(last_loss) > abs(loss) else lr * 0.5
            lr = min(lr, 1.0)
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)
        last_loss = loss
        mt.shuffle(n, perm)       # base/iterativeRecommender.py:101
    assert np.array_equal(np.stack([u0[perm], i0[perm]], 1), z[f"order{len(meta['epochs'])}"])
    assert np.array_equal(mt.words625(), z["py_state"])
    # test-set predictions: P[u].Q[i] clipped to the rating scale and round(.,3)
    # (base/recommender.py:88-110, base/iterativeRecommender.py:65-73)
    tu, ti = z["test_uid"], z["test_iid"]
    ok = (tu >= 0) & (ti >= 0)
    pred = np.einsum("nd,nd->n", P[tu[ok]], Q[ti[ok]])
    lo, hi = meta["rScale"][0], meta["rScale"][-1]
    want = z["test_pred"][ok]
    got = np.where(pred > hi, hi, np.where(pred < lo, lo, np.round(pred, 3)))
    np.testing.assert_allclose(got, want, atol=1.01e-3)  # round() boundary cases may flip the last digit


def _bold_driver(lr, last_loss, loss, epoch):
    """base/iterativeRecommender.py:59-63,91-99 (learning-rate update inside isConverged)"""
    if not abs(last_loss - loss) < 1e-3:
        if epoch > 1:
            lr = lr * 1.05 if abs(last_loss) > abs(loss) else lr * 0.5
        lr = min(lr, 1.0)
    return lr


@pytest.mark.parametrize("name,variant", [("pmf_filmtrust", 1), ("svd_filmtrust", 2), ("ee_filmtrust", 3)])
def test_pmf_svd_filmtrust(golden_dir, name, variant):
    """model/rating/PMF.py:9-28 and model/rating/SVD.py:13-35 through the oracle, against the recorded
    reference runs (tables, biases, losses, learning-rate schedule, shuffle stream, predictions)."""
    meta, z = _load(golden_dir, name)
    P, Q = z["P0"].copy(), z["Q0"].copy()
    Bu = z["Bu0"].copy() if variant >= 2 else None
    Bi = z["Bi0"].copy() if variant >= 2 else None
    regU, regI, regB, gm = meta["regU"], meta["regI"], meta["regB"], meta["globalMean"]
    mt = O.MT.cpython_seed(meta["seed"])
    n = z["order0"].shape[0]
    perm = np.arange(n, dtype=np.int64)
    u0 = np.ascontiguousarray(z["order0"][:, 0]); i0 = np.ascontiguousarray(z["order0"][:, 1]); r0 = z["rating0"]
    lr = meta["epochs"][0]["lr_used"]; last_loss = 0.0
    for k, ep in enumerate(meta["epochs"]):
        u = np.ascontiguousarray(u0[perm]); i = np.ascontiguousarray(i0[perm]); r = np.ascontiguousarray(r0[perm])
        assert np.array_equal(np.stack([u, i], 1), z[f"order{k}"])
        assert lr == pytest.approx(ep["lr_used"], rel=1e-15)
        loss = O.mf_sgd_variant(variant, P, Q, u, i, r, lr, regU, regI, Bu, Bi, regB, gm)
        if variant != 3:
            loss += regU * O.sumsq(P) + regI * O.sumsq(Q)
        if variant == 2:
            loss += regB * (O.sumsq(Bu) + O.sumsq(Bi))
        if variant == 3:
            loss += regB * O.sumsq(Bu) + regB * O.sumsq(Bi)
        assert loss == pytest.approx(ep["loss"], rel=1e-11)
        np.testing.assert_allclose(P, z[f"P{k+1}"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(Q, z[f"Q{k+1}"], rtol=1e-10, atol=1e-13)
        if variant >= 2:
            np.testing.assert_allclose(Bu, z[f"Bu{k+1}"], rtol=1e-10, atol=1e-13)
            np.testing.assert_allclose(Bi, z[f"Bi{k+1}"], rtol=1e-10, atol=1e-13)
        lr = _bold_driver(lr, last_loss, loss, ep["epoch"])
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)
        last_loss = loss
        mt.shuffle(n, perm)
    assert np.array_equal(mt.words625(), z["py_state"])
    tu, ti = z["test_uid"], z["test_iid"]
    ok = (tu >= 0) & (ti >= 0)
    pred = np.einsum("nd,nd->n", P[tu[ok]], Q[ti[ok]])
    if variant == 2:
        pred = pred + gm + Bi[ti[ok]] + Bu[tu[ok]]
    if variant == 3:
        diff = P[tu[ok]] - Q[ti[ok]]
        pred = gm + Bi[ti[ok]] + Bu[tu[ok]] - np.einsum("nd,nd->n", diff, diff)
    lo, hi = meta["rScale"][0], meta["rScale"][-1]
    got = np.where(pred > hi, hi, np.where(pred < lo, lo, np.round(pred, 3)))
    np.testing.assert_allclose(got, z["test_pred"][ok], atol=1.01e-3)


def test_pairwise_sampler_stream(golden_dir):
    """base/deepRecommender.py:29-52 over two epochs: shuffle(trainingData) then one
    negative per row, membership against ALL train items of the user."""
    meta, z = _load(golden_dir, "pairwise_adj_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    uid, iid = z["train_uid"], z["train_iid"]
    rated = user_item_csr(uid, iid, np.ones(uid.size), U, I).sorted_rows()
    mt = O.MT.cpython_seed(meta["seed"])
    perm = np.arange(uid.size, dtype=np.int64)
    out = []
    for ep in range(meta["epochs_sampled"]):
        mt.shuffle(uid.size, perm)
        ru = np.ascontiguousarray(uid[perm]); ri = np.ascontiguousarray(iid[perm])
        neg = O.pairwise_sample_epoch(mt, ru, rated.indptr, rated.indices, I)
        out.append(np.stack([ru, ri, neg], 1))
    stream = np.concatenate(out)
    assert np.array_equal(stream, z["stream"])
    assert np.array_equal(mt.words625(), z["py_state"])
    bs = z["batch_sizes"]; B = meta["batch_size"]
    assert (bs[bs != B] < B).all() and bs.sum() == stream.shape[0]


def test_find_k_largest_on_reference_reclists(golden_dir):
    """util/qmath.py:134-146 + mask-to-0 (base/recommender.py:147-149) reproduce the
    reference's recommendation lists from the reference's final P,Q."""
    meta, z = _load(golden_dir, "bpr_filmtrust")
    last = len(meta["epochs"])
    P, Q = z[f"P{last}"], z[f"Q{last}"]
    U, I = meta["n_users"], meta["n_items"]
    rated = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I)
    N = z["rec_ids"].shape[1]
    for row, u in enumerate(z["rec_users"][:300]):
        if u < 0:
            continue
        cand = Q.dot(P[u])
        cand[rated.indices[rated.indptr[u]:rated.indptr[u + 1]]] = 0
        ids, sc = O.find_k_largest(N, cand)
        assert np.array_equal(ids, z["rec_ids"][row])
        np.testing.assert_allclose(sc, z["rec_scores"][row], rtol=1e-12)


def test_svdpp_filmtrust(golden_dir):
    """model/rating/SVDPlusPlus.py:25-86 through the oracle against the recorded reference run: P, Q, Y, biases,
    losses, learning-rate schedule, shuffle stream."""
    meta, z = _load(golden_dir, "svdpp_filmtrust")
    P, Q, Y, Bu, Bi = (z[k].copy() for k in ("P0", "Q0", "Y0", "Bu0", "Bi0"))
    regU, regI, regB, regY, gm = meta["regU"], meta["regI"], meta["regB"], meta["regY"], meta["globalMean"]
    U, I = meta["n_users"], meta["n_items"]
    u0 = np.ascontiguousarray(z["order0"][:, 0]); i0 = np.ascontiguousarray(z["order0"][:, 1]); r0 = z["rating0"]
    rated = user_item_csr(u0, i0, r0, U, I)           # trainSet_u in dict order == userRated()
    mt = O.MT.cpython_seed(meta["seed"])
    n = u0.size
    perm = np.arange(n, dtype=np.int64)
    lr = meta["epochs"][0]["lr_used"]; last_loss = 0.0
    for k, ep in enumerate(meta["epochs"]):
        u = np.ascontiguousarray(u0[perm]); i = np.ascontiguousarray(i0[perm]); r = np.ascontiguousarray(r0[perm])
        assert np.array_equal(np.stack([u, i], 1), z[f"order{k}"])
        loss = O.svdpp_sgd(P, Q, Y, Bu, Bi, rated.indptr, rated.indices, u, i, r, lr, regU, regI, regB, regY, gm)
        loss += regU * O.sumsq(P) + regI * O.sumsq(Q) + regY * O.sumsq(Y) + regB * (O.sumsq(Bu) + O.sumsq(Bi))
        assert loss == pytest.approx(ep["loss"], rel=1e-11)
        for got, name in ((P, "P"), (Q, "Q"), (Y, "Y"), (Bu, "Bu"), (Bi, "Bi")):
            np.testing.assert_allclose(got, z[f"{name}{k+1}"], rtol=1e-10, atol=1e-13)
        lr = _bold_driver(lr, last_loss, loss, ep["epoch"])
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)
        last_loss = loss
        mt.shuffle(n, perm)
    assert np.array_equal(mt.words625(), z["py_state"])


def test_sept_graph_builders_match_the_reference(golden_dir):
    """model/ranking/SEPT.py:32-114 (pure scipy + random.sample in the reference): friend / sharing views, the joint
    adjacency and two sampled sub-adjacencies recorded from the live reference; the restatement must reproduce them
    (structure exactly, values to the last bit in the reference's dtype) with the CPython stream replayed by the oracle."""
    import scipy.sparse as sp
    from oracle import tfmodels as T
    meta, z = _load(golden_dir, "sept_graphs_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    uid, iid, fo, fe = z["train_uid"], z["train_iid"], z["follower"], z["followee"]

    def same(A, tag, dtype):
        A = sp.csr_matrix(A); A.sort_indices()
        assert np.array_equal(A.indptr, z[tag + "_indptr"]) and np.array_equal(A.indices, z[tag + "_indices"]), tag
        assert A.data.dtype == dtype and np.array_equal(A.data.astype(np.float64), z[tag + "_data"]), tag
    social, sharing = T.sept_social_views(U, I, uid, iid, fo, fe)
    same(social, "social", np.float64); same(sharing, "sharing", np.float64)
    same(T.sept_sub_adjacency(U, I, uid, iid, fo, fe), "full", np.float32)
    m = O.MT.from_python_state((3, tuple(int(x) for x in z["state_before_sub1"]), None))
    for tag in ("sub1", "sub2"):
        assert np.array_equal(m.words625(), z[f"state_before_{tag}"])
        keep = m.sample_range(uid.size, int(uid.size * (1 - meta["drop_rate"])))        # SEPT.py:86
        skeep = m.sample_range(fo.size, int(fo.size * (1 - meta["drop_rate"])))         # SEPT.py:92
        same(T.sept_sub_adjacency(U, I, uid, iid, fo, fe, keep, skeep), tag, np.float32)
    assert np.array_equal(m.words625(), z["state_after"])
    assert meta["relations_kept"] == fo.size < meta["relations_loaded"]


def test_tbpr_filmtrust_stream_tables_and_loss(golden_dir):
    """model/ranking/TBPR.py (numpy path) through the oracle against the recorded run of the unmodified reference: the
    chained triplet stream bit-exact (joint / weak / strong item lists as that process ordered them), P and Q after
    every epoch, the loss with its per-user regularisation terms, the learning-rate schedule, the generator state."""
    meta, z = _load(golden_dir, "tbpr_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    P, Q = z["P0"].copy(), z["Q0"].copy()
    pos = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I, min_rating=1)
    sets = [(z[t + "_indptr"], z[t + "_items"]) for t in ("joint", "weak", "strong")]
    mt = O.MT.cpython_seed(meta["seed"])
    lr, last, streams = meta["epochs"][0]["lr_used"], 0.0, []
    for ep in meta["epochs"]:
        u, a, b = O.tbpr_sample_epoch(mt, pos.indptr, pos.indices, I, *sets)
        assert u.size == meta["triplets_per_epoch"]
        loss = O.tbpr_epoch(P, Q, u, a, b, lr, meta["regU"], meta["regI"])
        k = ep["epoch"]
        np.testing.assert_allclose(P, z[f"P{k}"], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(Q, z[f"Q{k}"], rtol=1e-11, atol=1e-14)
        assert loss == pytest.approx(ep["loss"], rel=1e-12) and lr == pytest.approx(ep["lr_used"], rel=1e-15)
        if not abs(last - loss) < 1e-3:                     # base/iterativeRecommender.py:88-104
            if k > 1:
                lr = lr * 1.05 if abs(last) > abs(loss) else lr * 0.5
            lr = min(lr, 1.0)
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)
        last = loss
        mt.shuffle(meta["n_train"])
        streams.append(np.stack([u, a, b], axis=1))
    stream = np.concatenate(streams)
    assert np.array_equal(stream, z["steps"]) and _sha(stream) == meta["stream_sha256"]
    assert np.array_equal(mt.words625()[:624], z["py_state"][:624]) and mt.words625()[624] == z["py_state"][624]
    assert (stream[:, 1] == stream[:, 2]).any()             # the last draw does repeat a social item now and then: rows alias


def _sbpr_inputs(meta, z):
    """what the SBPR loop reads, rebuilt from the fixture: PositiveSet CSR, FPSet CSR + counts, the item-name -> user-id links of
    `item_j in self.FPSet`, the keys FPSet has after initModel"""
    U, I = meta["n_users"], meta["n_items"]
    pos = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], U, I, min_rating=1)
    name2user = {n: k for k, n in enumerate(z["user_names"].tolist())}
    link = np.array([name2user.get(n, -1) for n in z["item_names"].tolist()], dtype=np.int32)
    is_key = (np.diff(z["fp_indptr"]) > 0).astype(np.uint8)
    return pos, link, is_key


def test_sbpr_filmtrust_stream_tables_and_loss(golden_dir):
    """model/ranking/SBPR.py (numpy path) through the oracle's restatement (oracle/npref.py) against the recorded run of the
    reference's source with the ONE token of line 46 replaced (the file as it is raises TypeError there -- the fixture records that
    too): the (u, i, k, j, Suk) rows bit-exact, P and Q after every epoch, the loss, the learning-rate schedule, the generator."""
    import random
    from oracle import npref
    meta, z = _load(golden_dir, "sbpr_filmtrust")
    raised = meta["unmodified_reference_raises"]
    assert raised["type"] == "TypeError" and raised["line"] == 46 and "kItems" in raised["statement"]
    pos, link, is_key = _sbpr_inputs(meta, z)
    assert (link >= 0).sum() > 1000                        # FilmTrust: most item names ARE user names -- the :52 test bites
    rng = random.Random(); rng.seed(meta["seed"])
    P, Q, b = z["P0"].copy(), z["Q0"].copy(), z["b"]
    per = z["stream"].shape[0] // len(meta["epochs"])
    lr, last = meta["epochs"][0]["lr_used"], 0.0
    for ep in meta["epochs"]:
        k = ep["epoch"]
        rows = npref.sbpr_sample_epoch(rng, z["positive_set_users"], pos.indptr, pos.indices, z["fp_indptr"], z["fp_items"], z["fp_counts"],
                                       meta["n_items"], link, is_key)
        assert np.array_equal(rows, z["stream"][(k - 1) * per:k * per])
        loss = npref.sbpr_epoch(P, Q, b, z["positive_set_users"], rows, lr, meta["regU"], meta["regI"])
        np.testing.assert_allclose(P, z[f"P{k}"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(Q, z[f"Q{k}"], rtol=1e-12, atol=1e-15)
        assert loss == pytest.approx(ep["loss"], rel=1e-13) and lr == pytest.approx(ep["lr_used"], rel=1e-15)
        if not abs(last - loss) < 1e-3:                     # base/iterativeRecommender.py:88-104
            if k > 1:
                lr = lr * 1.05 if abs(last) > abs(loss) else lr * 0.5
            lr = min(lr, 1.0)
        assert lr == pytest.approx(ep["lr_next"], rel=1e-15)
        last = loss
        order = list(range(meta["n_train"])); rng.shuffle(order)        # isConverged: shuffle(trainingData)
    assert np.array_equal(np.array(rng.getstate()[1], dtype=np.uint32), z["py_state"])
    st = z["stream"]
    assert _sha(st) == meta["stream_sha256"] and (st[:, 2] >= 0).mean() > 0.3 and ((st[:, 2] == st[:, 3]) & (st[:, 2] >= 0)).any()


def test_mhcn_graph_builders_match_the_reference(golden_dir):
    """model/ranking/MHCN.py:26-85 (pure scipy / python in the reference): the three motif-induced channel adjacencies
    and the value list of buildJointAdjacency, bit for bit in the reference's float32."""
    from oracle import tfmodels as T
    meta, z = _load(golden_dir, "mhcn_graphs_filmtrust")
    U, I = meta["n_users"], meta["n_items"]
    H = T.mhcn_motif_adjacencies(U, I, z["train_uid"], z["train_iid"], z["follower"], z["followee"])
    for tag, A, nnz in zip(("Hs", "Hj", "Hp"), H, meta["nnz"]):
        A.sort_indices()
        assert A.nnz == nnz and A.data.dtype == np.float32
        assert np.array_equal(A.indptr, z[tag + "_indptr"]) and np.array_equal(A.indices, z[tag + "_indices"]) and np.array_equal(A.data, z[tag + "_data"])
    R = T.mhcn_joint_adjacency(U, I, z["train_uid"], z["train_iid"], z["train_r"]).tocoo()
    want = {}
    for (u, i), v in zip(z["R_indices"].tolist(), z["R_values"].tolist()):       # a training row listed twice contributes twice
        want[(u, i)] = np.float32(want.get((u, i), np.float32(0)) + np.float32(v))
    assert R.shape == tuple(meta["R_shape"]) and R.nnz == len(want) < z["R_values"].size
    assert all(want[(int(u), int(i))] == v for u, i, v in zip(R.row, R.col, R.data))


def test_numpy_level_restatement_of_the_reference_loop_equals_the_c_restatement():
    """oracle/npref.py (what bench.py times as the interpreter-bound CPU form, model/ranking/BPR.py:28-53) against the C oracle:
    the same negatives from the same generator state, tables and loss to rounding; a bounded sample stops where it is told to."""
    import random
    from oracle import npref
    from qrec_amd.synth import make_dataset, to_csr
    d = make_dataset("small"); U, I = d["n_users"], d["n_items"]
    indptr, ind = to_csr(U, d["train_u"], d["train_i"])
    u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
    rng = np.random.default_rng(0); P0 = rng.random((U, 10)) / 3; Q0 = rng.random((I, 10)) / 3
    P, Q = P0.copy(), Q0.copy()
    nll, done, negs = npref.bpr_epoch(P, Q, indptr, ind, I, 0.05, 0.01, 0.02, rng=random.Random(5))
    assert done == ind.size
    j = O.bpr_sample_epoch(O.MT.from_python_state(random.Random(5).getstate()), indptr, ind, I)
    assert np.array_equal(j, negs)
    Pc, Qc = P0.copy(), Q0.copy()
    lref = O.bpr_sgd(Pc, Qc, u, ind, j, 0.05, 0.01, 0.02)
    assert abs(nll - lref) / lref < 1e-12 and np.abs(P - Pc).max() < 1e-12 and np.abs(Q - Qc).max() < 1e-12
    _, part, negs2 = npref.bpr_epoch(P0.copy(), Q0.copy(), indptr, ind, I, 0.05, 0.01, 0.02, rng=random.Random(5), max_triplets=1000)
    assert part == 1000 and np.array_equal(negs2, negs[:1000])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_golden_generator_reproduces_the_committed_fixtures_byte_for_byte(tmp_path):
    """tests/golden/README.md's claim, checked: the generator re-run against the unchanged reference writes the committed bytes.  Two
    cases in a few seconds: the pointwise sampler, and TBPR -- the one run whose course depends on the interpreter's string-hash seed
    (TBPR.py:122 iterates a set of item names), which the generator pins to PYTHONHASHSEED=0 and records."""
    import hashlib
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, QREC_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONHASHSEED", None)
    subprocess.run([sys.executable, os.path.join(here, "gen_golden.py"), "case_tbpr_filmtrust", "case_pointwise"], check=True, env=env,
                   capture_output=True, timeout=600)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    for name in ("tbpr_filmtrust.npz", "pointwise_filmtrust.npz"):
        assert sha(os.path.join(str(tmp_path), name)) == sha(os.path.join(here, name)), name
    new = json.load(open(os.path.join(str(tmp_path), "golden_meta.json"))); old = json.load(open(os.path.join(here, "golden_meta.json")))
    assert new["tbpr_filmtrust"] == old["tbpr_filmtrust"] and new["pointwise_filmtrust"] == old["pointwise_filmtrust"]
    assert old["tbpr_filmtrust"]["python_hash_seed"] == "0"
