#!/usr/bin/env python3
"""Paired Recall@20 harness: every throughput mode that is NOT order-exact against order-exact fp64 training.

north_star: "Recall@20 within +-0.002 of reference".  The metric is util/measure.py:106-109 over the lists of
base/recommender.py:127-179; the training it is measured after is model/ranking/BPR.py:19-53 with the bold driver of
base/iterativeRecommender.py:56-63.  Paired design: the GPU run and the reference run start from the same tables, use the SAME
negative for every (u, i) of every epoch (the device Philox stream, read back), and the same learning-rate rule -- what differs is
the execution: fp32 + Hogwild staleness + the schedule's reordering (+ for N > 1 the layout's cross-rank staleness) against the
reference's strictly sequential fp64 order (oracle/qrec_oracle.c, the checker; nothing here is on a timed path).

Datasets -- ground on which the bar CAN fail (the structureless Zipf graph peaks at Recall@20 0.034):
  yelp2018-clustered   synth.gen_edges_clustered, the Yelp2018 shape with 64 planted communities: order-exact training reaches
                       Recall@20 ~0.12;
  lastfm               the reference's own dataset/lastfm split as recorded in tests/golden/bpr_lastfm.npz (1,888 x 15,314,
                       74,272 train / 18,562 test rows): the reference reaches 0.099 after two epochs (golden_meta.json);
  yelp2018             the structureless graph of the bench line, for continuity.

Modes: "item" (item-major, the default; P[u] by atomic deltas), "user", "item:rmw" / "item:auto" (P[u] by sc1 load + store / chosen by the
collision density: engine.resolve_p_update; round 6).  (The two-pass "item-deferred" modes of rounds 3-5 are gone with their kernels; result
files of those rounds under profiles/ still name them, and `reference_from` reads such files for the reference curve of the same stored order.)
N > 1: G logical ranks in this process (threads + tests/logical_ranks.ThreadComm, the real kernels and exchange code), layouts
"replicated" (users sharded, item table replicated, per-epoch delta all-reduce) and "sharded" (item table row-sharded, per-batch
row exchange) -- bench.py's layouts.

usage: paired_recall.py <out.json> [plan]      plan = "full" (default) | "quick" | "bpr-conf" | JSON list of cases (or a file holding one)
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REG, MAX_LR, DIM, FLUSH = 0.001, 1.0, 64, 16          # config/BPR.conf:9-10 (-u 0.001 -i 0.001, -max 1)
BAR = 0.002


def load_dataset(name: str) -> dict:
    from qrec_amd.synth import make_dataset, to_csr
    if name == "lastfm":
        z = np.load(os.path.join(ROOT, "tests", "golden", "bpr_lastfm.npz"))
        tu, ti = z["train_uid"].astype(np.int32), z["train_iid"].astype(np.int32)
        U, I = int(tu.max()) + 1, int(ti.max()) + 1
        order = np.argsort(tu, kind="stable")                 # PositiveSet order: users by id, a user's items in row order
        ok = (z["test_uid"] >= 0) & (z["test_iid"] >= 0)       # test rows whose user / item the training file never saw cannot be hit
        d = dict(n_users=U, n_items=I, train_u=tu[order], train_i=ti[order], test_u=z["test_uid"][ok].astype(np.int32),
                 test_i=z["test_iid"][ok].astype(np.int32), shape="lastfm")
    else:
        d = make_dataset(name)
    d["indptr"], d["items"] = to_csr(d["n_users"], d["train_u"], d["train_i"])
    d["u"] = np.repeat(np.arange(d["n_users"], dtype=np.int32), np.diff(d["indptr"])).astype(np.int32)
    return d


def initial_tables(d: dict, seed: int = 3, dim: int = DIM):
    rng = np.random.default_rng(seed)                      # rand/3, base/iterativeRecommender.py:37-38
    return (rng.random((d["n_users"], dim)) / 3).astype(np.float32), (rng.random((d["n_items"], dim)) / 3).astype(np.float32)


def recall20(P, Q, d, N: int = 20) -> float:
    import bench as B
    return B.evaluate_recall(P, Q, d, d["indptr"], d["items"], N=N)


def parse_mode(mode: str):
    """"item[:rmw|:auto|:atomic]" / "user" -> (schedule, P[u] update policy).  Modes of rounds 3-5's result files ("item-deferred:4:fresh")
    parse to their stored order: ("item", "atomic")."""
    parts = mode.split(":")
    schedule = parts[0].split("-")[0]
    p_update = parts[1] if len(parts) > 1 and parts[1] in ("rmw", "auto", "atomic") else "atomic"
    return schedule, p_update


def _rank_problem(d, world, rank):
    from qrec_amd import dist as qd
    if world == 1:
        return 0, d["n_users"], d["indptr"], d["items"], d["u"]
    lo, hi, lp, li = qd.shard_positive_csr(d["indptr"], d["items"], world, rank)
    return lo, hi, lp, li, np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(lp)).astype(np.int32)


def n_batches_for(d, world, layout, shard_batch, syncs):
    """batches per epoch, the same on every rank: sharded -- no rank's batch above ``shard_batch`` triplets, two at least from 2^19
    triplets (dist.agree_on_batches); replicated -- ``syncs`` reconciliations of the item table per epoch"""
    if world == 1:
        return 1
    if layout == "sharded":
        n_max = max(int(_rank_problem(d, world, r)[3].size) for r in range(world))
        nb = max(1, -(-n_max // shard_batch), int(syncs))
        return max(nb, 2) if n_max >= (1 << 19) else nb
    return max(1, int(syncs))


def build_rank(d, mode, world, rank, layout, P0, Q0, shard_batch=1 << 20, syncs=1, item_run=None, rounds=None):
    """(tables, sgd, chunk, lo, hi, groups) of one rank: its users' rows of P, the item table whole (replicated) or its row shard.
    ``rounds``: engine.grid_for_epoch's min_rounds (None = the engine's default, 0 / 1 = the launcher's own grid, rounds 1-4)"""
    from qrec_amd import dist as qd
    from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk
    from qrec_amd.interactions import CSR
    schedule, p_update = parse_mode(mode)
    lo, hi, lp, li, lu = _rank_problem(d, world, rank)
    sharded = world > 1 and layout == "sharded"
    t = DeviceTables(P0[lo:hi], qd.shard_item_rows(Q0, world, rank) if sharded else Q0, np.float32)
    chunk = balanced_chunk(int(li.size))
    batches = n_batches_for(d, world, layout, shard_batch, syncs)
    sgd = BprSgd(t, lu, li, CSR(lp, li), schedule=schedule, n_items=d["n_items"], batches=batches,
                 chunk=chunk, item_run=item_run, p_update=p_update)
    chunk, groups = sgd.launch_grid(rounds)                 # (an epoch in batches: engine.launch_chunk per batch, unchanged)
    return t, sgd, chunk, lo, hi, groups


def train_rank(d, sgd, t, chunk, lr0, seed, epochs, marks, world, rank, comm, layout, on_mark, stream=None, groups=0):
    """``epochs`` epochs of one rank in throughput mode as bench.py's epoch runs them: device sampler, SGD kernel(s), the layout's
    collectives, device-side epoch close with the bold driver.  ``on_mark(epoch, P_local, Q_local)`` at the epochs in ``marks``
    (every rank calls it).  Returns the device driver's log."""
    from qrec_amd import capi, dist as qd
    sharded = world > 1 and layout == "sharded"
    dstep = None
    if sharded:
        dstep = qd.ShardedStep(comm, qd.ShardedItemExchange(comm, d["n_items"], t.ld, t.Q), len(sgd.batch_bounds) - 1)
    elif world > 1:
        dstep = qd.ReplicatedStep(comm, qd.ReplicatedTableSync(comm, t.Q))
    sgd.start_device_driver(lr0, log_capacity=epochs)
    capi.device_sync()
    for k in range(epochs):
        sgd.sample_negatives_device(seed + 7919 * rank, k, stream)
        if sharded:
            dstep.prepare(sgd, stream)
        sgd.epoch_device_async(REG, REG, MAX_LR, tol=0.0, chunk=chunk, flush_every=FLUSH, stream=stream, dist=dstep, groups=groups)
        if k + 1 in marks:
            capi.device_sync()
            Pr, Qr = t.download(np.float32)
            on_mark(k + 1, Pr, Qr)
    capi.device_sync()
    return sgd.driver_log(stream)


def assemble(P_parts, Q_parts, world, layout, n_items):
    """whole tables from the ranks' pieces: user blocks in rank order; item rows interleaved (sharded) or rank 0's copy (replicated)"""
    P = np.concatenate(P_parts) if world > 1 else P_parts[0]
    if world > 1 and layout == "sharded":
        Q = np.empty((n_items, P.shape[1]), np.float32)
        for r in range(world):
            Q[r::world] = Q_parts[r][:len(range(r, n_items, world))]
        return P, Q
    return P, Q_parts[0]


def negatives_of(samplers, seed):
    """k -> epoch k's negatives of all ranks in the reference's triplet order (user blocks are contiguous: rank order)"""
    def negatives(k):
        out = []
        for r, sgd in enumerate(samplers):
            sgd.sample_negatives_device(seed + 7919 * r, k)
            out.append(sgd.negatives_reference_order())
        return np.concatenate(out)
    return negatives


def gpu_run(d, mode, lr0, seed, epochs, marks, world=1, layout="replicated", P0=None, Q0=None, shard_batch=1 << 20, syncs=1, item_run=None,
            rounds=None, extra_topn=()):
    """world = 1, or G logical ranks in this process.  Returns {recall: {mark: r}, loss: [...], lr: [...], negatives: k -> j}."""
    from qrec_amd import capi
    if world > 1:        # G logical ranks: the in-process test transport (never needed by bench.py's N = 1 legs)
        from tests.logical_ranks import ThreadComm, run_ranks
    state = {"recall": {}, "P": [None] * world, "Q": [None] * world, "sgd": [None] * world, "log": None}

    def rank_main(rank, group):
        t, sgd, chunk, lo, hi, groups = build_rank(d, mode, world, rank, layout, P0, Q0, shard_batch, syncs, item_run, rounds)
        state["sgd"][rank] = sgd
        state["grid"] = (chunk, groups)

        def on_mark(epoch, Pr, Qr):
            state["P"][rank], state["Q"][rank] = Pr, Qr
            if world > 1:
                group.barrier.wait()
            if rank == 0:
                PQ = assemble(state["P"], state["Q"], world, layout, d["n_items"])
                state["recall"][epoch] = recall20(*PQ, d)
                if epoch == epochs:
                    state["final_topn"] = {int(N): recall20(*PQ, d, N=int(N)) for N in extra_topn}
            if world > 1:
                group.barrier.wait()

        log = train_rank(d, sgd, t, chunk, lr0, seed, epochs, marks, world, rank, ThreadComm(group, rank) if world > 1 else None, layout,
                         on_mark, capi.Stream() if world > 1 else None, groups=groups)
        if rank == 0:
            state["log"] = log

    if world == 1:
        capi.init(0)
        rank_main(0, None)
    else:
        run_ranks(world, rank_main)
    log = state["log"]
    return {"recall": state["recall"], "loss": [float(x) for x in log[:, 0]], "lr": [float(x) for x in log[:, 1]], "grid": state.get("grid"),
            "final_topn": state.get("final_topn", {}),
            "negatives": negatives_of(state["sgd"], seed), "samplers": state["sgd"],
            "perm_key": (parse_mode(mode)[0], world, layout if world > 1 else "", len(state["sgd"][0].batch_bounds), state["sgd"][0].item_run),
            "p_update": state["sgd"][0].p_update, "collision_density": state["sgd"][0].collision}


def reference_run(d, negatives, lr0, epochs, marks, P0, Q0, order=None, extra_topn=()):
    """order-exact fp64 training (oracle C restatement of BPR.py:45-53,40) on the same negatives, bold driver of
    iterativeRecommender.py:56-63 (isConverged's threshold is not applied: every epoch runs, as on the GPU side with tol = 0).
    ``order``: visit the same triplets sequentially in ANOTHER order (indices into the reference's order) -- not the reference any
    more, but the sequential statement of a schedule's own visiting order: what is left between it and the GPU run is the parallel
    execution (Hogwild staleness, fp32), what is left between it and the reference is the order itself."""
    from oracle import c as O
    P, Q = P0.astype(np.float64), Q0.astype(np.float64)
    lr, last, rec, losses, lrs = lr0, 0.0, {}, [], []
    uu, ii = (d["u"], d["items"]) if order is None else (np.ascontiguousarray(d["u"][order]), np.ascontiguousarray(d["items"][order]))
    for k in range(epochs):
        j = negatives(k) if order is None else np.ascontiguousarray(negatives(k)[order])
        lrs.append(lr)
        loss = O.bpr_sgd(P, Q, uu, ii, j, lr, REG, REG) + REG * O.sumsq(P) + REG * O.sumsq(Q)
        losses.append(float(loss))
        if k > 0:
            lr *= 1.05 if abs(last) > abs(loss) else 0.5
        lr = min(lr, MAX_LR); last = loss
        if k + 1 in marks:
            rec[k + 1] = recall20(P, Q, d)
    return {"recall": rec, "loss": losses, "lr": lrs, "final_topn": {int(N): recall20(P, Q, d, N=int(N)) for N in extra_topn}}


def item_major_visit_order(sgd, chunk: int) -> np.ndarray:
    """the one-pass item-major kernel's time order as a sequence: time slot s runs chunk (s * stride) mod n_chunks of the item-sorted
    list, stride ~ 0.618 n_chunks made coprime (csrc/bpr_sgd.hip launch_hogwild_item); the groups' interleaving inside a round
    of slots is not modelled"""
    import math
    n = sgd.n
    n_chunks = -(-n // chunk)
    stride = max(int(n_chunks * 0.6180339887498949), 1)
    while math.gcd(stride, n_chunks) != 1:
        stride += 1
    slots = (np.arange(n_chunks, dtype=np.int64) * stride) % n_chunks
    at = (slots[:, None] * chunk + np.arange(chunk)[None, :]).ravel()
    return sgd.perm[at[at < n]]


def cached_reference(path: str, case: dict, perm_key):
    """The reference side of an EARLIER run of this harness (a results file under profiles/), for a case whose reference costs minutes of
    the GPU box's time (25 M triplets x 30 epochs of sequential fp64: 7 min): the reference run is a function of (dataset, seed, rate, epochs,
    dimension, stored order) alone -- the negatives are the counter-based device stream, the tables and the dataset are seeded -- so a case
    of that file with the same values and a mode of the same stored order holds the same curve.  Recall at the marks only (the file keeps
    the loss at three marks and no learning rates): the result says so and leaves the loss gap / bold-driver comparison empty."""
    same = ("dataset", "lr0", "seed", "epochs")
    perm_schedule, world, layout, n_bounds, item_run = perm_key
    #  the negatives -- and so the reference curve -- depend on the per-rank seeds and on the stored order: only a single-rank,
    # single-batch case may take a cached curve, and only from a case of the same stored order
    if case.get("world", 1) != 1 or world != 1 or n_bounds != 2:
        raise KeyError(f"reference_from needs a single-rank, single-batch case (got world {case.get('world', 1)}, {n_bounds - 1} batches)")
    for c in json.load(open(os.path.join(ROOT, path)))["cases"]:
        if "curve" not in c or any(c.get(k) != case.get(k) for k in same) or c.get("eval_every", 5) != case.get("eval_every", 5):
            continue
        if c.get("dim", DIM) != case.get("dim", DIM) or c.get("init_seed", 3) != case.get("init_seed", 3) or c.get("world", 1) != 1:
            continue
        if parse_mode(c["mode"])[0] != perm_schedule or c.get("item_run") != case.get("item_run") or c.get("layout", "replicated") != case.get("layout", "replicated"):
            continue
        return {"recall": {int(m): float(r) for m, _, r in c["curve"]}, "loss": None, "lr": None, "final_topn": {},
                "cached_from": f"{path}: case mode {c['mode']!r} (recall at the marks only)"}
    raise KeyError(f"{path} holds no case with the reference of {case}")


def compare(case: dict, g: dict, r: dict) -> dict:
    marks = sorted(r["recall"])
    peak = max(marks, key=lambda m: r["recall"][m])
    final = marks[-1]
    have_log = r.get("loss") is not None

    def at(m):
        a, b = g["recall"][m], r["recall"][m]
        return {"epoch": m, "recall_gpu": a, "recall_exact_order": b, "abs_diff": abs(a - b), "rel_diff": abs(a - b) / max(b, 1e-12),
                "signed_diff": a - b, "loss_gpu": g["loss"][m - 1], "loss_exact_order": r["loss"][m - 1] if have_log else None,
                "loss_rel_gap": abs(g["loss"][m - 1] - r["loss"][m - 1]) / abs(r["loss"][m - 1]) if have_log else None}
    worst = max(marks, key=lambda m: abs(g["recall"][m] - r["recall"][m]))
    same_lr = bool(np.allclose(g["lr"], r["lr"], rtol=1e-9)) if have_log else None
    topn = {str(N): {"recall_gpu": g["final_topn"][N], "recall_exact_order": r["final_topn"][N], "signed_diff": g["final_topn"][N] - r["final_topn"][N]}
            for N in g.get("final_topn", {}) if N in r.get("final_topn", {})}
    return {**case, "peak": at(peak), "final": at(final), "worst_mark": at(worst), "bar": BAR, "grid_chunk_groups": g.get("grid"),
            **({"final_other_topn": topn} if topn else {}),
            "within_bar_at_peak": abs(g["recall"][peak] - r["recall"][peak]) <= BAR,
            "within_bar_at_every_mark": abs(g["recall"][worst] - r["recall"][worst]) <= BAR,
            "same_bold_driver_decisions": same_lr,
            "curve": [[m, g["recall"][m], r["recall"][m]] for m in marks]}


def run_case(case: dict, cache: dict, datasets: dict) -> dict:
    name, lr0, seed, mode = case["dataset"], case["lr0"], case["seed"], case["mode"]
    world, layout = case.get("world", 1), case.get("layout", "replicated")
    epochs, every = case["epochs"], case.get("eval_every", 5)
    marks = set(range(every, epochs + 1, every)) | {epochs}
    if name not in datasets:
        datasets[name] = load_dataset(name)
    d = datasets[name]
    P0, Q0 = initial_tables(d, case.get("init_seed", 3), case.get("dim", DIM))
    topn = tuple(case.get("extra_topn", ()))
    t0 = time.perf_counter()
    from qrec_amd.dist import reconciliations_per_epoch
    g = gpu_run(d, mode, lr0, seed, epochs, marks, world, layout, P0, Q0, shard_batch=case.get("shard_batch", 1 << 20),
                syncs=reconciliations_per_epoch(world, case.get("syncs", 0)),
                item_run=case.get("item_run"), rounds=case.get("rounds"), extra_topn=topn)
    t1 = time.perf_counter()
    key = (name, lr0, seed, epochs, every, case.get("init_seed", 3), case.get("dim", DIM), topn) + g["perm_key"]
    if key not in cache:      # the negatives are a function of (seed, epoch, stored order): modes with the same order share a reference
        if case.get("reference_from"):
            cache[key] = cached_reference(case["reference_from"], case, g["perm_key"])
        else:
            cache[key] = reference_run(d, g["negatives"], lr0, epochs, marks, P0, Q0, extra_topn=topn)
    out = compare({k: v for k, v in case.items()}, g, cache[key])
    if cache[key].get("cached_from"):
        out["reference_side"] = cache[key]["cached_from"]
    else:           # kept whole, so that a later run can take this reference from the results file
        out["reference_log"] = {"loss": cache[key]["loss"], "lr": cache[key]["lr"]}
    out["gpu_log"] = {"loss": g["loss"], "lr": g["lr"]}
    out["p_update"], out["collision_density"] = g["p_update"], g["collision_density"]
    if case.get("order_null"):
        # the yardstick (no GPU involved): the SAME sequential fp64 training on the same negatives with the epoch's triplets visited in one
        # fixed random order instead of the reference's -- how far the reference's own measure moves under a reordering at this setting
        nkey = key + ("null",)
        if nkey not in cache:
            order = np.random.default_rng(1_000_003 + seed).permutation(d["items"].size)
            cache[nkey] = reference_run(d, g["negatives"], lr0, epochs, marks, P0, Q0, order=order)
        fin = max(cache[key]["recall"])
        out["order_null"] = {"what": "sequential fp64, one fixed random visiting order, minus sequential fp64 in the reference's order (final epoch)",
                             "signed_diff": cache[nkey]["recall"][fin] - cache[key]["recall"][fin]}
    if case.get("own_order") and world == 1 and mode == "item":
        from qrec_amd.engine import balanced_chunk
        okey = key + ("own",)
        if okey not in cache:
            cache[okey] = reference_run(d, g["negatives"], lr0, epochs, marks, P0, Q0, order=item_major_visit_order(g["samplers"][0], balanced_chunk(d["items"].size)))
        own, ref = cache[okey], cache[key]
        vs_own = compare({}, g, own)
        out["vs_sequential_in_own_order"] = {k: vs_own[k] for k in ("peak", "final", "worst_mark")}
        out["order_effect_alone"] = {"what": "sequential fp64 in the schedule's order vs sequential fp64 in the reference's order (no GPU involved)",
                                     "curve": [[m, own["recall"][m], ref["recall"][m]] for m in sorted(ref["recall"])],
                                     "max_abs_diff": max(abs(own["recall"][m] - ref["recall"][m]) for m in ref["recall"])}
    out["seconds"] = {"gpu_side": t1 - t0, "reference_side": time.perf_counter() - t1}
    return out


def plan_full():
    """the ledger of profiles/r04_paired_recall.json: every mode that is not order-exact, on both structured datasets, at BPR.conf's rate
    and five times it; N = 2 / 4 (/ 8) logical ranks in both layouts; the 6 M-triplet regime `auto` switches schedules in"""
    Y, L, X = "yelp2018-clustered", "lastfm", "xl6m-clustered"
    runs = {Y: ((0.01, 40, 5), (0.05, 20, 5)), L: ((0.01, 40, 4), (0.05, 20, 2))}
    cases = []
    for ds in (Y, L):
        for lr0, epochs, every in runs[ds]:
            for seed in (7, 11):
                for mode in ("item", "user"):
                    cases.append(dict(dataset=ds, lr0=lr0, seed=seed, mode=mode, epochs=epochs, eval_every=every))
            cases.append(dict(dataset=ds, lr0=lr0, seed=7, mode="item", epochs=epochs, eval_every=every, item_run=0, own_order=True))
    for ds in (Y, L):
        for lr0, epochs, every in runs[ds]:
            for world in (2, 4):
                for layout in ("replicated", "sharded"):
                    cases.append(dict(dataset=ds, lr0=lr0, seed=7, mode="item", epochs=epochs, eval_every=every, world=world, layout=layout,
                                      **({"shard_batch": 1 << 14} if ds == L and layout == "sharded" else {})))
    for world in (2, 4):
        cases.append(dict(dataset=Y, lr0=0.05, seed=7, mode="item", epochs=20, eval_every=5, world=world, layout="replicated", syncs=4))
    for layout in ("replicated", "sharded"):
        cases.append(dict(dataset=Y, lr0=0.01, seed=7, mode="item", epochs=40, eval_every=5, world=8, layout=layout))
    for lr0, epochs, every in ((0.05, 15, 3), (0.01, 40, 5)):
        for mode in ("item", "item:rmw"):
            cases.append(dict(dataset=X, lr0=lr0, seed=7, mode=mode, epochs=epochs, eval_every=every))
    return cases


def plan_bpr_conf(seeds=range(1, 17), rounds=(0, 8, 32), modes=("item",), epochs=100, dataset="lastfm"):
    """The reference's OWN BPR workload (round 5): config/BPR.conf -- lastfm, num.factors 50, learnRate 0.01 (-max 1),
    reg 0.001, 100 epochs -- scored where the reference scores it: once, after the LAST epoch (model/ranking/BPR.py:28-43 ->
    base/recommender.py:181-212).  One run per seed (the seed draws the initial tables AND the negatives); Recall@20 (north_star) and
    Recall@10 (the conf's -topN 10) at the final epoch.  ``rounds``: engine.grid_for_epoch's minimum rounds of the grid (0 = rounds 1-4:
    the whole 74 k-triplet epoch in flight at once)."""
    cases = []
    for seed in seeds:
        for mode in modes:
            for r in rounds:
                cases.append(dict(dataset=dataset, lr0=0.01, seed=int(seed), init_seed=int(seed), dim=50, mode=mode, epochs=epochs, eval_every=10,
                                  **({} if r is None else {"rounds": int(r)}), extra_topn=[10], order_null=True))      # (rounds None: the engine's default)
    return cases


def summarize_seeds(results, keys=("dataset", "lr0", "mode", "rounds", "world", "layout", "syncs", "dim", "epochs")):
    """groups of cases that differ only in the seed -> signed final-epoch gap (GPU - reference): n, mean, sd, se, mean |gap|, max |gap|;
    same for the order-only yardstick where it was run"""
    groups = {}
    for r in results:
        if "final" in r:
            groups.setdefault(tuple((k, r.get(k)) for k in keys if r.get(k) is not None), []).append(r)
    out = []
    for k, rs in groups.items():
        if len(rs) < 2:
            continue

        def stats(x):
            x = np.asarray(x, dtype=np.float64)
            return {"n": int(x.size), "mean_signed": float(x.mean()), "sd": float(x.std(ddof=1)), "se": float(x.std(ddof=1) / np.sqrt(x.size)),
                    "mean_abs": float(np.abs(x).mean()), "max_abs": float(np.abs(x).max())}
        row = dict(k)
        row["seeds"] = [r["seed"] for r in rs]
        row["recall_exact_order_mean"] = float(np.mean([r["final"]["recall_exact_order"] for r in rs]))
        row["recall_exact_order_sd_over_seeds"] = float(np.std([r["final"]["recall_exact_order"] for r in rs], ddof=1))
        row["final_gap"] = stats([r["final"]["signed_diff"] for r in rs])
        row["peak_gap"] = stats([r["peak"]["signed_diff"] for r in rs])
        gaps = [r["final"]["loss_rel_gap"] for r in rs if r["final"]["loss_rel_gap"] is not None]
        row["final_loss_rel_gap_mean"] = float(np.mean(gaps)) if gaps else None
        row["same_bold_driver_decisions"] = int(sum(bool(r["same_bold_driver_decisions"]) for r in rs))
        tn = [r["final_other_topn"] for r in rs if "final_other_topn" in r]
        if tn:
            row["final_gap_other_topn"] = {N: stats([t[N]["signed_diff"] for t in tn]) for N in tn[0]}
        nulls = [r["order_null"]["signed_diff"] for r in rs if "order_null" in r]
        if nulls:
            row["order_null_final_gap"] = stats(nulls)
        out.append(row)
    return out


def plan_quick():
    return [dict(dataset="lastfm", lr0=0.05, seed=7, mode=m, epochs=10, eval_every=5) for m in ("item", "item:rmw")] + \
           [dict(dataset="lastfm", lr0=0.05, seed=7, mode="item", epochs=10, eval_every=5, world=2, layout=l, shard_batch=1 << 14)
            for l in ("replicated", "sharded")]


def main():
    out_path = sys.argv[1]
    plan = sys.argv[2] if len(sys.argv) > 2 else "full"
    cases = (plan_full() if plan == "full" else plan_quick() if plan == "quick" else plan_bpr_conf() if plan == "bpr-conf"
             else json.loads(open(plan).read() if os.path.exists(plan) else plan))
    from qrec_amd import capi
    capi.init(0)
    cache, datasets, results = {}, {}, []
    for c in cases:
        try:
            res = run_case(c, cache, datasets)
        except Exception as e:      # noqa: BLE001 -- a failing case must not lose the others' results
            res = {**c, "error": repr(e)}
        results.append(res)
        brief = {k: res.get(k) for k in ("dataset", "lr0", "seed", "mode", "world", "layout", "syncs", "item_run", "rounds") if res.get(k) is not None}
        if "peak" in res:
            brief.update(peak_epoch=res["peak"]["epoch"], recall=round(res["peak"]["recall_exact_order"], 5), abs_diff=round(res["peak"]["abs_diff"], 5),
                         rel=round(res["peak"]["rel_diff"], 4), worst=round(res["worst_mark"]["abs_diff"], 5), final=round(res["final"]["abs_diff"], 5),
                         loss_gap=None if res["final"]["loss_rel_gap"] is None else round(res["final"]["loss_rel_gap"], 4))
        else:
            brief["error"] = res["error"]
        print(json.dumps(brief), flush=True)
        json.dump({"_what": __doc__.split("\n\n")[0], "bar": BAR, "over_seeds": summarize_seeds(results), "cases": results}, open(out_path, "w"), indent=1)
    for row in summarize_seeds(results):
        print("over seeds:", json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
