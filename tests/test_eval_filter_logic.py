"""The decision logic of the fused evaluation's bf16 route, modelled in numpy and attacked with adversarial rounding errors.

The GPU kernels (eval_topk.hip: sample_max_bf16 / threshold_var / score_filter_bf16 / select_topk) implement these steps for a
user with exact (fp32) scores s_i, approximate (bf16) scores s^_i, |s^_i - s_i| <= eps, rated items masked out:
  1. group maxima of s^ over <= 128 item groups; tau^ = the (N+1)-th largest group maximum whose best item is unrated;
  2. tau = tau^ - eps (N+1 distinct unrated items have s >= tau);  tau_low = tau - eps;  tau <= 0 -> the exact walk;
  3. candidates = unrated items with s^ >= tau_low;
  4. L = the (N+1)-th largest s^ among the candidates; survivors = candidates with s^ >= L - 2 eps;
  5. survivors are re-scored exactly; the N+1 best by (score, lower id first); equal scores among them -> the exact walk.
Claim: whenever the user is not sent to the exact walk, the N best are exactly the N best unrated items.  The model below runs
the five steps with errors chosen to hurt (true top items pushed down by eps, their challengers pushed up by eps, random signs
elsewhere) and must never lose an item.  It tests the reasoning, not the kernels -- those are held to the block route and to the
reference's lists on the GPU (tests/test_gpu_eval.py)."""
import numpy as np
import pytest


def fused_route_model(s, s_hat, eps, rated, N, n_groups=64):
    n = s.size
    M = N + 1
    groups = np.array_split(np.arange(n), n_groups)
    gmax = np.array([s_hat[g].max() for g in groups]); garg = np.array([g[np.argmax(s_hat[g])] for g in groups])
    ok = ~np.isin(garg, rated)
    if ok.sum() < M:
        return None
    tau_hat = np.sort(gmax[ok])[-M]
    tau = tau_hat - eps
    if not tau > 0:
        return None
    tau_low = tau - eps
    cand = np.flatnonzero(s_hat >= tau_low)
    cand = cand[~np.isin(cand, rated)]
    if cand.size < M:
        return None
    L = np.sort(s_hat[cand])[-M]
    surv = cand[s_hat[cand] >= L - 2 * eps * 1.01]
    order = np.lexsort((surv, -s[surv]))              # exact score descending, lower id first
    top = surv[order][:M]
    if np.unique(s[top]).size < top.size:
        return None                                   # a tie among the N + 1 best: the heap's history decides, exact walk
    return top[:N]


@pytest.mark.parametrize("seed", range(12))
def test_no_item_is_lost_under_adversarial_rounding(seed):
    rng = np.random.default_rng(seed)
    n, N = 4096, 20
    decided = 0
    for trial in range(40):
        s = rng.standard_normal(n) * 0.1 + 0.6 if trial % 2 else rng.random(n) * 0.3 + 0.4          # dense near the top in the second form
        eps = float(rng.choice([1e-4, 1e-3, 5e-3, 2e-2]))
        rated = rng.choice(n, rng.integers(0, 60), replace=False)
        if trial % 5 == 0:
            rated = np.union1d(rated, np.argsort(-s)[:rng.integers(1, 15)])                           # the best items are rated ones
        unrated = np.setdiff1d(np.arange(n), rated)
        truth = unrated[np.lexsort((unrated, -s[unrated]))][:N]
        e = rng.uniform(-eps, eps, n)
        e[truth] = -eps                                                                               # push the true list down ...
        chal = unrated[np.lexsort((unrated, -s[unrated]))][N:N + 200]
        e[chal] = eps                                                                                 # ... and its challengers up
        if trial % 3 == 0:
            e[rated] = eps                                                                            # rated items look as good as they can
        got = fused_route_model(s, s + e, eps, rated, N, n_groups=int(rng.choice([32, 64, 119])))
        if got is None:
            continue
        decided += 1
        assert np.array_equal(got, truth), (trial, eps)
    assert decided >= 20          # the model decides most users itself; the rest go to the exact walk


def test_threshold_is_reached_by_enough_unrated_items_even_when_rated_items_lead_their_groups():
    rng = np.random.default_rng(99)
    n, N, eps = 2048, 10, 1e-3
    s = rng.random(n)
    rated = np.argsort(-s)[:40]                       # the forty best are rated: their groups are discarded
    e = rng.uniform(-eps, eps, n)
    got = fused_route_model(s, s + e, eps, rated, N, n_groups=128)
    unrated = np.setdiff1d(np.arange(n), rated)
    assert got is not None and np.array_equal(got, unrated[np.argsort(-s[unrated])][:N])
