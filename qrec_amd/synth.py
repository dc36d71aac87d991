"""Deterministic synthetic interaction matrices (SURVEY.md §8d "synthetic generator").

The reference ships no ML-1M / Yelp2018 data (/root/reference/.MISSING_LARGE_BLOBS), so the
bench and the parity tests use matrices of the named *shapes*, generated here:

* users / items get Zipf-like popularity p_k ∝ k^-alpha (alpha_user=0.4, alpha_item=0.6),
  randomly permuted;
* one edge per user and per item first (coverage), then 1.25·E random (u,i) pairs,
  de-duplicated keeping the first occurrence, truncated to E;
* rating = 1; rows are emitted **user-major in first-appearance order**, the order the
  reference's ``Rating`` data model assigns ids in (data/rating.py:48-54), so row r of the
  CSR is user id r and item ids are first-appearance ids as well.

Everything is numpy on the host; nothing here is on the timed path.
"""
from __future__ import annotations

import numpy as np

SHAPES = {
    # name: (users, items, train_edges, test_edges, seed)
    "ml1m": (6040, 3706, 1000209, 0, 1000),
    "yelp2018": (31668, 38048, 1237259, 324147, 2018),
    # same sizes, but 64 planted communities: 80 % of a user's interactions fall on items of the user's own community
    # (gen_edges_clustered) -- a graph WITH locality to harvest, next to the structureless one (SpMM L2 work)
    "yelp2018-clustered": (31668, 38048, 1237259, 324147, 2018),
    # a planted-community graph with a 6 M-triplet epoch (160 k users: collision density 0.03 -- P[u] stays with atomic deltas there,
    # engine.resolve_p_update; the fidelity test of the choice `auto` makes at this size runs on it)
    "xl6m-clustered": (160000, 100000, 6000000, 1500000, 6006),
    # ... and one at the size the HBM-resident roofline figure is quoted on (25 M triplets per epoch; with d = 128 its tables are 0.54 GB)
    "xl25m-clustered": (650000, 400000, 25000000, 6250000, 2525),
    # a second graph on which `auto` writes P[u] by load + store (1 M users: collision density 0.005), small enough for several paired runs
    "xl12m-clustered": (1000000, 200000, 12000000, 3000000, 1212),
    "tiny": (300, 200, 6000, 1500, 7),
    "small": (2000, 1500, 60000, 15000, 11),
}


def _zipf_probs(n: int, alpha: float, rng: np.random.Generator) -> np.ndarray:
    p = np.arange(1, n + 1, dtype=np.float64) ** (-alpha)
    p /= p.sum()
    return p[rng.permutation(n)]


def gen_edges(n_users: int, n_items: int, n_edges: int, seed: int,
              alpha_user: float = 0.4, alpha_item: float = 0.6):
    """Return (u, i) int64 arrays of exactly ``n_edges`` distinct pairs covering every
    user and every item at least once (needs n_edges >= n_users + n_items)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pu = _zipf_probs(n_users, alpha_user, rng)
    pi = _zipf_probs(n_items, alpha_item, rng)
    # coverage edges
    cu = np.arange(n_users, dtype=np.int64)
    ci = rng.choice(n_items, size=n_users, p=pi)
    du = rng.choice(n_users, size=n_items, p=pu)
    di = np.arange(n_items, dtype=np.int64)
    m = int(1.25 * n_edges)
    ru = rng.choice(n_users, size=m, p=pu)
    ri = rng.choice(n_items, size=m, p=pi)
    u = np.concatenate([cu, du, ru])
    i = np.concatenate([ci, di, ri])
    key = u * np.int64(n_items) + i
    _, first = np.unique(key, return_index=True)
    first.sort()
    if first.size < n_edges:
        raise ValueError("not enough distinct pairs; raise the oversampling factor")
    first = first[:n_edges]
    return u[first], i[first]


def gen_edges_clustered(n_users: int, n_items: int, n_edges: int, seed: int, n_clusters: int = 64, p_in: float = 0.8,
                        alpha_user: float = 0.4, alpha_item: float = 0.6):
    """As ``gen_edges`` but with planted communities: users and items are assigned to ``n_clusters`` communities at
    random; a draw picks the user by popularity and, with probability ``p_in``, an item OF THE USER'S COMMUNITY by
    (within-community) popularity, otherwise any item by popularity.  Ids stay randomly permuted, so the locality is
    in the graph, not in the numbering."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pu = _zipf_probs(n_users, alpha_user, rng)
    pi = _zipf_probs(n_items, alpha_item, rng)
    cl_u = rng.integers(0, n_clusters, n_users)
    cl_i = rng.integers(0, n_clusters, n_items)
    by_cl = np.argsort(cl_i, kind="stable")                         # items grouped by community
    start = np.searchsorted(cl_i[by_cl], np.arange(n_clusters + 1))
    w = pi[by_cl]
    tot = np.add.reduceat(w, start[:-1])
    cdf = np.cumsum(w) - np.repeat(np.concatenate([[0.0], np.cumsum(tot)[:-1]]), np.diff(start))
    cdf = cdf / np.repeat(tot, np.diff(start)) + np.repeat(np.arange(n_clusters), np.diff(start))   # community c: (c, c+1]

    def draw_items(users):
        glob = rng.choice(n_items, size=users.size, p=pi)
        c = cl_u[users]
        k = np.searchsorted(cdf, c + rng.random(users.size) * (1 - 1e-12), side="left")
        k = np.clip(k, start[c], start[c + 1] - 1)
        return np.where(rng.random(users.size) < p_in, by_cl[k], glob)

    cu = np.arange(n_users, dtype=np.int64)
    ci = draw_items(cu)
    du = rng.choice(n_users, size=n_items, p=pu)
    di = np.arange(n_items, dtype=np.int64)
    m = int(1.4 * n_edges)
    ru = rng.choice(n_users, size=m, p=pu)
    ri = draw_items(ru)
    u = np.concatenate([cu, du, ru])
    i = np.concatenate([ci, di, ri])
    key = u * np.int64(n_items) + i
    _, first = np.unique(key, return_index=True)
    first.sort()
    if first.size < n_edges:
        raise ValueError("not enough distinct pairs; raise the oversampling factor")
    first = first[:n_edges]
    return u[first], i[first]


def relabel_first_appearance(u: np.ndarray, i: np.ndarray):
    """Relabel users/items by first appearance while walking the rows in order — exactly
    the id assignment of the reference data model (data/rating.py:48-54)."""
    def relabel(x):
        _, first_idx, inv = np.unique(x, return_index=True, return_inverse=True)
        order = np.argsort(first_idx, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        return rank[inv].astype(np.int64)
    return relabel(u), relabel(i)


def make_dataset(shape: str = "yelp2018", *, test_fraction: float | None = None):
    """Build train/test COO arrays for a named shape.

    Returns dict(n_users, n_items, train_u, train_i, test_u, test_i).  Train rows are
    user-major (grouped by user, users in first-appearance order); ids are
    first-appearance ids *of the train file*, as the reference would assign them.
    The test split takes the last ``test_fraction`` of each user's edges (Yelp2018 ratio
    324147/1561406 by default when the shape has test edges).
    """
    n_users, n_items, e_train, e_test, seed = SHAPES[shape]
    total = e_train + e_test
    u, i = (gen_edges_clustered if shape.endswith("-clustered") else gen_edges)(n_users, n_items, total, seed)
    # group by user, stable (keeps each user's draw order)
    order = np.argsort(u, kind="stable")
    u, i = u[order], i[order]
    if e_test > 0:
        frac = e_test / total if test_fraction is None else test_fraction
        counts = np.bincount(u, minlength=n_users)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        pos = np.arange(total) - np.repeat(starts, counts)
        n_test_u = np.floor(counts * frac).astype(np.int64)
        n_test_u = np.minimum(n_test_u, counts - 1)  # keep >=1 train edge per user
        is_test = pos >= np.repeat(counts - n_test_u, counts)
    else:
        is_test = np.zeros(total, dtype=bool)
    tr_u, tr_i = u[~is_test], i[~is_test]
    te_u, te_i = u[is_test], i[is_test]
    # the reference assigns ids walking the TRAIN file; items only seen in test keep no id
    # (they can never be recommended).  Relabel train, map test through the same tables.
    _, fu, inv_u = np.unique(tr_u, return_index=True, return_inverse=True)
    ou = np.argsort(fu, kind="stable"); ru = np.empty_like(ou); ru[ou] = np.arange(ou.size)
    uniq_i, fi, inv_i = np.unique(tr_i, return_index=True, return_inverse=True)
    oi = np.argsort(fi, kind="stable"); ri = np.empty_like(oi); ri[oi] = np.arange(oi.size)
    train_u = ru[inv_u].astype(np.int32)
    train_i = ri[inv_i].astype(np.int32)
    uniq_u = np.unique(tr_u)
    # map test ids; drop test rows whose item never appears in train
    pos_i = np.searchsorted(uniq_i, te_i)
    pos_i = np.clip(pos_i, 0, uniq_i.size - 1)
    ok = uniq_i[pos_i] == te_i
    pos_u = np.searchsorted(uniq_u, te_u)
    test_u = ru[pos_u[ok]].astype(np.int32)
    test_i = ri[pos_i[ok]].astype(np.int32)
    return dict(n_users=int(uniq_u.size), n_items=int(uniq_i.size),
                train_u=train_u, train_i=train_i, test_u=test_u, test_i=test_i,
                shape=shape, seed=seed)


def to_csr(n_rows: int, rows: np.ndarray, cols: np.ndarray):
    """Row-grouped COO -> (indptr int64, indices int32), keeping within-row order."""
    order = np.argsort(rows, kind="stable")
    counts = np.bincount(rows, minlength=n_rows)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, cols[order].astype(np.int32)


def write_rating_file(path: str, u: np.ndarray, i: np.ndarray) -> None:
    """Emit the reference's on-disk format, one ``"user item 1"`` row per line
    (read by util/io.py:31-76).  Names are the integer ids prefixed so that user and
    item name spaces cannot be confused."""
    with open(path, "w") as f:
        f.write("".join(f"u{a} i{b} 1\n" for a, b in zip(u.tolist(), i.tolist())))
