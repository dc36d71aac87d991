"""GPU parity tests of full-rank evaluation (MFMA scoring + mask-to-0 + heap top-N)."""
import numpy as np
import pytest

from oracle import c as O
from qrec_amd import capi
from qrec_amd.interactions import CSR, user_item_csr
from qrec_amd.ranking import DeviceRanker

from helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    capi.init(0)
    yield


def _oracle_topk(U, V, rated, users, N):
    ids = np.empty((len(users), N), np.int32); sc = np.empty((len(users), N), np.float64)
    for r, u in enumerate(users):
        cand = (V.astype(np.float64) @ U[u].astype(np.float64)) if U.dtype == np.float64 else (V @ U[u]).astype(np.float64)
        if rated is not None:
            cand[rated.indices[rated.indptr[u]:rated.indptr[u + 1]]] = 0
        i, s = O.find_k_largest(N, cand)
        ids[r, :i.size] = i; sc[r, :s.size] = s
    return ids, sc


def test_reference_reclists_fp64():
    """The reference's own recommendation lists (FilmTrust run) from its final P, Q."""
    meta, z = load_golden("bpr_filmtrust")
    last = len(meta["epochs"])
    P, Q = z[f"P{last}"], z[f"Q{last}"]
    rated = user_item_csr(z["train_uid"], z["train_iid"], z["train_r"], meta["n_users"], meta["n_items"])
    ok = z["rec_users"] >= 0
    users = z["rec_users"][ok]
    ids, sc = DeviceRanker(P, Q, rated).topk(users, z["rec_ids"].shape[1])
    assert np.array_equal(ids, z["rec_ids"][ok])
    np.testing.assert_allclose(sc, z["rec_scores"][ok], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(70, 33, 8), (300, 1000, 64), (129, 257, 50), (64, 40, 128), (40, 20, 200)])
def test_ties_negative_scores_and_masked_zeros(dtype, shape):
    """Small-integer embeddings: scores are exact integers in every precision, so ties are
    everywhere, many scores are negative (masked zeros outrank them) -- the id lists must
    still be bit-identical to the reference's heap procedure."""
    n_users, n_items, d = shape
    rng = np.random.default_rng(n_users * 7 + d)
    U = rng.integers(-2, 3, (n_users, d)).astype(dtype); V = rng.integers(-2, 3, (n_items, d)).astype(dtype)
    uu = rng.integers(0, n_users, 4 * n_users); ii = rng.integers(0, n_items, 4 * n_users)
    rated = user_item_csr(uu, ii, np.ones(uu.size), n_users, n_items)
    users = np.arange(n_users, dtype=np.int32)
    for N in (1, 10, 20, 100):
        k = min(N, n_items)
        ids, sc = DeviceRanker(U, V, rated).topk(users, k)
        oi, os_ = _oracle_topk(U, V, rated, users, k)
        assert np.array_equal(ids, oi), (dtype, shape, N)
        assert np.array_equal(sc.astype(np.float64), os_)


def test_random_embeddings_fp32_within_tolerance_and_no_mask():
    rng = np.random.default_rng(5)
    U = rng.standard_normal((500, 64)).astype(np.float32); V = rng.standard_normal((3000, 64)).astype(np.float32)
    users = rng.permutation(500)[:333].astype(np.int32)       # ragged batch, arbitrary order
    ids, sc = DeviceRanker(U, V, None).topk(users, 20)
    ref = V.astype(np.float64) @ U[users].astype(np.float64).T           # [items, users]
    for r in range(users.size):
        col = ref[:, r]
        np.testing.assert_allclose(sc[r], col[ids[r]], rtol=1e-5, atol=1e-5)   # scores are the items' scores
        assert (np.diff(sc[r]) <= 0).all()
        kth = np.sort(col)[-20]
        assert (col[ids[r]] >= kth - 1e-4).all()                               # and they are the top 20
    with pytest.raises(ValueError):
        DeviceRanker(U, V, None).topk(np.array([500], np.int32), 5)
    assert DeviceRanker(U, V, None).topk(np.zeros(0, np.int32), 5)[0].shape == (0, 5)


def test_yelp_shape_full_eval_properties():
    """Eval at the bench shape: 31,668 users x 38,048 items, d=64, N=20 (154 GFLOP)."""
    from qrec_amd.synth import make_dataset, to_csr
    d = make_dataset("yelp2018")
    U_, I_ = d["n_users"], d["n_items"]
    rng = np.random.default_rng(1)
    U = (rng.random((U_, 64)) / 3 - 0.1).astype(np.float32); V = (rng.random((I_, 64)) / 3 - 0.1).astype(np.float32)
    indptr, ind = to_csr(U_, d["train_u"], d["train_i"])
    rated = CSR(indptr, ind)
    users = np.arange(U_, dtype=np.int32)
    ids, sc = DeviceRanker(U, V, rated).topk(users, 20)
    assert ids.shape == (U_, 20) and (ids >= 0).all() and (ids < I_).all()
    assert (np.diff(sc, axis=1) <= 0).all()
    # spot-check 200 users against the oracle procedure on fp32 scores recomputed in fp64
    pick = rng.permutation(U_)[:200]
    for u in pick:
        col = V.astype(np.float64) @ U[u].astype(np.float64)
        col[ind[indptr[u]:indptr[u + 1]]] = 0
        np.testing.assert_allclose(sc[u], col[ids[u]], rtol=2e-5, atol=2e-6)
        assert (col[ids[u]] >= np.sort(col)[-20] - 1e-5).all()
    # a rated item can only appear with score exactly 0
    for u in pick[:50]:
        r = set(ind[indptr[u]:indptr[u + 1]].tolist())
        for i, s in zip(ids[u], sc[u]):
            assert (i not in r) or s == 0.0


def test_rank_hits_kernel_equals_python_measures():
    """qrec_rank_hits on device-resident lists vs Measure.hits / Measure.NDCG (util/measure.py:15-21,70-82)
    evaluated on the same lists: hit counts equal, DCG sums bit-identical, for several list lengths."""
    import math
    rng = np.random.default_rng(5)
    U_, I_, d, N = 700, 900, 16, 20
    U = rng.standard_normal((U_, d)).astype(np.float32); V = rng.standard_normal((I_, d)).astype(np.float32)
    tu = rng.integers(0, U_, 6000).astype(np.int32); ti = rng.integers(0, I_, 6000).astype(np.int32)
    test = user_item_csr(tu, ti, np.ones(tu.size), U_, I_)
    rk = DeviceRanker(U, V, None); rk.set_test(test)
    users = rng.permutation(U_)[:650].astype(np.int32)
    ids, sc, per_cut = rk.topk(users, N, cuts=[1, 7, 20])
    srt = test.sorted_rows()
    for c in (1, 7, 20):
        hits, dcg = per_cut[c]
        for b, u in enumerate(users.tolist()):
            truth = set(srt.indices[srt.indptr[u]:srt.indptr[u + 1]].tolist())
            want_h = len(truth.intersection(ids[b, :c].tolist()))
            want_d = sum(1.0 / math.log(pos + 2) for pos, it in enumerate(ids[b, :c].tolist()) if it in truth)
            assert hits[b] == want_h and dcg[b] == want_d
    _, _, again = rk.topk(users, N, cuts=[7], want_lists=False)
    assert np.array_equal(again[7][0], per_cut[7][0]) and np.array_equal(again[7][1], per_cut[7][1])


def test_fast_measures_equal_rankingMeasure_strings_incl_cold_users_and_unknown_items():
    """evalRanking's fast path (no result file wanted) must print exactly what Measure.rankingMeasure prints for
    the lists of rank_all_test_users: warm users on the device, cold users and unknown test items included."""
    import io
    from contextlib import redirect_stdout
    from qrec_amd.model.ranking.BPR import BPR
    from qrec_amd.util.measure import Measure
    from helpers import conf_from_text
    rng = np.random.default_rng(8)
    train = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(rng.integers(0, 300, 6000).tolist(), rng.integers(0, 400, 6000).tolist())]
    test = [[f"u{u}", f"i{i}", 1.0] for u, i in zip(rng.integers(0, 300, 1500).tolist(), rng.integers(0, 400, 1500).tolist())]
    test += [["cold_a", "i3", 1.0], ["cold_a", "i5", 1.0], ["cold_b", "never_seen", 1.0], ["u1", "never_seen", 1.0]]
    conf = conf_from_text("ratings=./x.txt\nmodel.name=BPR\nratings.setup=-columns 0 1 2\nevaluation.setup=-testSet x\nitem.ranking=on -topN 5,10,20\n"
                          "num.factors=16\nnum.max.epoch=2\nlearnRate=-init 0.05 -max 1\nreg.lambda=-u 0.01 -i 0.01 -b 0.2 -s 0.2\n"
                          "output.setup=off -dir ./results/")
    import random
    random.seed(1); np.random.seed(1)
    with redirect_stdout(io.StringIO()):
        m = BPR(conf, train, test)
        got = m.execute()                                      # fast path (output off)
        want = Measure.rankingMeasure(m.data.testSet_u, m.rank_all_test_users(20), [5, 10, 20])
        quick = m.ranking_performance(0)
    assert got == want
    assert quick == [x.strip() for x in want[11:]]             # in-training variant: top-max(N) only


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_tied_users", [0, 7, 96])
@pytest.mark.parametrize("N", [1, 20, 100])
def test_sliced_topk_and_its_exact_fallbacks(dtype, n_tied_users, N):
    """The sliced top-N (kernels 3b) on a catalogue long enough to be sliced, with exactly known scores: user k's
    embedding is the k-th unit vector, so score(i, k) = V[i][k].  Users with pairwise distinct scores take the fast
    path; users whose best N+1 scores contain ties are redone by the exact emulation -- a few of them through the
    wavefront-per-user kernel, all of them through the lane-per-user kernel -- and every list must equal the
    reference's heap procedure, ids included."""
    n_users, n_items = 96, 40_000
    rng = np.random.default_rng(N * 131 + n_tied_users)
    V = np.empty((n_items, n_users), dtype)
    for k in range(n_users):
        V[:, k] = rng.permutation(n_items) - n_items // 3                  # distinct, some negative
    tied = rng.permutation(n_users)[:n_tied_users]
    for k in tied:
        V[:, k] = rng.integers(-5, 40, n_items)                             # heavy ties, also inside the top N
    if n_tied_users:                                                        # one user whose ONLY tie is at the N / N+1 boundary
        k = int(tied[0]); col = rng.permutation(n_items).astype(dtype); order = np.argsort(-col)
        col[order[min(N, n_items - 1)]] = col[order[N - 1]]
        V[:, k] = col
    U = np.eye(n_users, dtype=dtype)
    uu = rng.integers(0, n_users, 500); ii = rng.integers(0, n_items, 500)
    rated = user_item_csr(uu, ii, np.ones(uu.size), n_users, n_items)
    users = rng.permutation(n_users).astype(np.int32)
    ids, sc = DeviceRanker(U, V, rated).topk(users, N)
    oi, os_ = _oracle_topk(U, V, rated, users, N)
    assert np.array_equal(ids, oi)
    assert np.array_equal(sc.astype(np.float64), os_)


def test_exact_walk_spans_several_lds_stages(monkeypatch):
    """45,000 items: the exact per-user walk of the fused route stages a row in LDS 39,936 items at a time -- the heap must
    carry over from one stage to the next.  A third of the users have only negative scores (threshold <= 0, so they take the
    walk), every 11th ties everywhere; lists equal to the block route's (which the other tests tie to the reference's procedure)."""
    rng = np.random.default_rng(45)
    n_users, n_items, d, N = 200, 45_000, 64, 20
    V = (rng.standard_normal((n_items, d)) * 0.3 + 0.2).astype(np.float32)
    V[40_100:40_140] = V[200:240]                                             # duplicates on both sides of the stage boundary
    U = (rng.standard_normal((n_users, d)) * 0.3 + 0.5).astype(np.float32)
    U[::3] = -np.abs(U[::3])
    U[5::11] = 0.0
    uu = rng.integers(0, n_users, 40 * n_users); ii = rng.integers(0, n_items, 40 * n_users)
    rated = user_item_csr(uu, ii, np.ones(uu.size), n_users, n_items)
    users = rng.permutation(n_users)[:187].astype(np.int32)
    ids_f, sc_f = DeviceRanker(U, V, rated).topk(users, N)
    monkeypatch.setenv("QREC_EVAL_BLOCK_PATH", "1")
    ids_b, sc_b = DeviceRanker(U, V, rated).topk(users, N)
    assert np.array_equal(ids_f, ids_b) and np.array_equal(sc_f, sc_b)


_FUSED_ROUTES = {
    "bf16": {},                                     # the default: bf16 threshold + filter passes, exact fp32 re-scoring
    "f32-filter": {"QREC_EVAL_F32_FILTER": "1"},    # both passes in fp32 (what d = 100 takes anyway)
    "bf16-sparse-sample": {"QREC_EVAL_BF16_STRIDE": "8"},   # long candidate lists: pool > 64, selection by wave-wide maxima
    "bf16-two-tiles": {"QREC_EVAL_NU": "2"},        # two user tiles per wavefront instead of four
}


@pytest.mark.parametrize("route", list(_FUSED_ROUTES))
@pytest.mark.parametrize("d,N", [(64, 20), (100, 10), (128, 63), (32, 1)])
def test_fused_route_equals_the_block_route(d, N, route, monkeypatch):
    """The fused evaluation (per-user threshold, score + filter without a users x items block, wave-per-user selection
    with exact fp32 re-scoring behind the bf16 filter, flagged users through the exact walk) against the block route
    (QREC_EVAL_BLOCK_PATH=1) on the same tables: identical ids and scores -- with item ids sorted by popularity (the best
    scores crowd the low ids), users whose scores are all negative (threshold <= 0: rated items, masked to 0, outrank
    everything), users with ties in their top N, and a ragged batch.  Every variant of the fused route is held to it."""
    for k, v in _FUSED_ROUTES[route].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(d * 1000 + N)
    n_users, n_items = 700, 20_000
    pop = (np.arange(n_items, dtype=np.float64) + 1) ** -0.5                  # item id ~ popularity rank
    V = (rng.standard_normal((n_items, d)) * 0.3 + pop[:, None] * 2.0).astype(np.float32)
    U = (rng.standard_normal((n_users, d)) * 0.3 + 0.5).astype(np.float32)
    U[::7] = -np.abs(U[::7])                                                  # every score of these users is negative
    U[3::50] = 0.0                                                            # all scores tie at 0
    V[5000:5040] = V[100:140]                                                 # exact duplicates: ties inside many top lists
    uu = rng.integers(0, n_users, 30 * n_users); ii = (rng.integers(0, n_items, 30 * n_users) ** 2 // n_items).astype(np.int64)
    rated = user_item_csr(uu, ii, np.ones(uu.size), n_users, n_items)
    users = rng.permutation(n_users)[:651].astype(np.int32)
    ids_f, sc_f = DeviceRanker(U, V, rated).topk(users, N)
    monkeypatch.setenv("QREC_EVAL_BLOCK_PATH", "1")
    ids_b, sc_b = DeviceRanker(U, V, rated).topk(users, N)
    assert np.array_equal(ids_f, ids_b) and np.array_equal(sc_f, sc_b)
    zero = np.isin(users, np.arange(3, n_users, 50))
    assert (sc_b[zero] == 0).all() and (sc_b[~zero][:, 0] != 0).any()


@pytest.mark.parametrize("route", ["bf16", "bf16-sparse-sample"])
@pytest.mark.parametrize("d,N", [(64, 20), (128, 10)])
def test_bf16_filter_is_complete_on_adversarial_tables(d, N, route, monkeypatch):
    """What the bf16 filter's bound (eval_topk.hip, eps = 2^-7 * 1.02 * |u| * max|v|) has to survive on the hardware, not in a
    model of it: item norms spread over six decades (eps is set by the LARGEST item, the deciding scores by small ones), users
    likewise, large cancelling coordinates (sum |u_k v_k| >> |u.v|: the rounding error is large against the score), and clusters
    of items whose fp32 scores sit within a few ulp of each other around every user's threshold.  A candidate the filter drops
    would change an id or a score against the block route; they have to be identical."""
    for k, v in _FUSED_ROUTES[route].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(17 * d + N)
    n_users, n_items = 300, 12_000
    V = rng.standard_normal((n_items, d))
    V *= np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n_items))[:, None]
    U = rng.standard_normal((n_users, d)) * np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n_users))[:, None]
    # cancellation: a huge +c, -c pair of coordinates on both sides
    U[:, 0] += 50.0 * np.abs(U).max(1); U[:, 1] = -U[:, 0] + rng.standard_normal(n_users) * 1e-3
    V[::5, 0] = 7.0; V[::5, 1] = 7.0
    # near-ties: 400 copies of a strong item, perturbed in the last bits of one coordinate
    base = np.abs(rng.standard_normal(d)) + 0.5
    for c in range(400):
        V[2000 + c] = base; V[2000 + c, 2 + c % (d - 2)] *= 1.0 + (c // 7) * 2.0 ** -22
    U32, V32 = U.astype(np.float32), V.astype(np.float32)
    uu = rng.integers(0, n_users, 20 * n_users); ii = rng.integers(0, n_items, 20 * n_users)
    rated = user_item_csr(uu, ii, np.ones(uu.size), n_users, n_items)
    users = np.arange(n_users, dtype=np.int32)
    ids_f, sc_f = DeviceRanker(U32, V32, rated).topk(users, N)
    monkeypatch.setenv("QREC_EVAL_BLOCK_PATH", "1")
    ids_b, sc_b = DeviceRanker(U32, V32, rated).topk(users, N)
    assert np.isfinite(sc_b).all()
    assert np.array_equal(ids_f, ids_b) and np.array_equal(sc_f, sc_b)
