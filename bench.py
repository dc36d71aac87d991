#!/usr/bin/env python3
"""bench.py -- BPR triplet-updates/sec on the MI355X hot path (BASELINE.json metric).

One *step* = one full pass of the reference's BPR epoch (model/ranking/BPR.py:29-43) over
the synthetic Yelp2018-shape interaction matrix (31,668 x 38,048, ~1.25 M train triplets,
d=64): negative sampling for every triplet (device Philox sampler) -> fused gather / dot /
sigmoid / SGD scatter kernel (throughput mode) -> epoch-end regulariser reductions ->
loss read-back -> bold-driver learning-rate update on the host (the reference's
isConverged, minus the data shuffle BPR never looks at).  Inputs are resident in HBM
before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1 (weak scaling): every rank owns its own population of 31,668 users (its rows of P
and its 1.25 M triplets per step); the item table Q is replicated and re-synchronised
every step by one all-reduce of the per-rank Q deltas over RCCL/xGMI.  `value` counts
the triplets of all ranks over the max-over-ranks time.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the SGD kernel's algorithmic bytes/launch over its mean launch time
                  (HIP events on the launch stream) against 8 TB/s HBM peak;
  cpu_baseline -- the CPU port of the same epoch (oracle/, plain C, fp64, 1 thread) timed
                  on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from qrec_amd import capi  # noqa: E402
from qrec_amd.capi import DeviceBuffer  # noqa: E402
from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk  # noqa: E402
from qrec_amd.interactions import CSR  # noqa: E402
from qrec_amd.synth import make_dataset, to_csr  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DIM = 64
LR0, MAX_LR, REG_U, REG_I = 0.01, 1.0, 0.001, 0.001   # config/BPR.conf:9-10
FLUSH_EVERY = 16


def bytes_per_triplet(d: int) -> int:
    """3 rows read + 3 rows written (fp32) + 3 int32 ids."""
    return 6 * d * 4 + 12


def recall_at_n(ids: np.ndarray, users: np.ndarray, test_u: np.ndarray, test_i: np.ndarray, n_items: int) -> float:
    """util/measure.py:106-109: mean over test users of |top-N ∩ test items| / |test items|."""
    test_keys = np.unique(test_u.astype(np.int64) * n_items + test_i)
    rec_keys = (users.astype(np.int64)[:, None] * n_items + ids).ravel()
    hit = np.isin(rec_keys, test_keys).reshape(ids.shape).sum(1)
    cnt = np.bincount(test_u, minlength=int(users.max()) + 1)[users]
    return float((hit / cnt).mean())


def evaluate_recall(P, Q, data, indptr, items, N=20):
    """Recall@N of (P, Q) on the held-out edges through the product's device ranker."""
    from qrec_amd.ranking import DeviceRanker
    users = np.unique(data["test_u"]).astype(np.int32)
    ids, _ = DeviceRanker(np.ascontiguousarray(P, dtype=np.float32), np.ascontiguousarray(Q, dtype=np.float32),
                          CSR(indptr, items)).topk(users, N)
    return recall_at_n(ids, users, data["test_u"], data["test_i"], data["n_items"])


def cpu_baseline(u, i, indptr, n_items, U, seconds=12.0):
    """The same epochs on the host, timed: oracle sampler (CPython MT19937 replay) + the plain-C fp64
    restatement of BPR.optimization + the epoch-end reductions, single thread, ~`seconds` of work."""
    from oracle import c as O
    rng = np.random.default_rng(0)
    P = rng.random((U, DIM)) / 3; Q = rng.random((n_items, DIM)) / 3
    mt = O.MT.cpython_seed(0)
    done, epochs, t0 = 0, 0, time.perf_counter()
    while True:
        j = O.bpr_sample_epoch(mt, indptr, i, n_items)
        O.bpr_sgd(P, Q, u, i, j, LR0, REG_U, REG_I)
        O.sumsq(P); O.sumsq(Q)
        done += u.size; epochs += 1
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "triplet-updates/s", "cores": 1, "kind": "port",
            "sample": f"{epochs} full epochs ({done} triplets, {dt:.1f} s) of the same Yelp2018-shape workload; "
                      "plain-C fp64 port of model/ranking/BPR.py:29-53 incl. the CPython-stream sampler "
                      "(the Python reference itself cannot travel to this box; it measured 58.9k/s on 1 core, BASELINE.md)"}


def cpu_exact_order_reference(sgd, u, i, n_items, P0, Q0, epochs, seed):
    """Recall@20 reference: order-exact fp64 training on the host for the same number of epochs, from
    the same initial tables, with the reference's bold-driver schedule and -- paired design -- the very
    negatives the GPU run used (the device Philox stream for (seed, epoch) is re-generated and read back)."""
    from oracle import c as O
    P, Q = P0.astype(np.float64), Q0.astype(np.float64)
    lr, last = LR0, 0.0
    for k in range(epochs):
        sgd.sample_negatives_device(seed, k)
        j = sgd.negatives_reference_order()          # same j for the same (u, i), whatever the GPU's visiting order
        loss = O.bpr_sgd(P, Q, u, i, j, lr, REG_U, REG_I) + REG_U * O.sumsq(P) + REG_I * O.sumsq(Q)
        if k > 0:
            lr *= 1.05 if abs(last) > abs(loss) else 0.5
        lr = min(lr, MAX_LR); last = loss
    return P, Q, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="yelp2018")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=capi.HW_DEFAULT)
    ap.add_argument("--flush-every", type=int, default=0, help="item-major schedule: triplets between flushes of the register-resident Q[i] (0 = default)")
    ap.add_argument("--schedule", choices=("item", "user"), default="item",
                    help="visiting order of the epoch's triplets in the Hogwild kernel (DESIGN.md s4)")
    args = ap.parse_args()

    # stdout carries ONE line, the result.  Libraries print there too -- RCCL writes its version banner to C stdio's
    # stdout, flushed at exit, i.e. AFTER a Python print -- so file descriptor 1 is pointed at stderr for the whole run and
    # the JSON line goes to the saved original descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dist = torch = None
    # QREC_FORCE_DIST=1 drives the torch.distributed/RCCL branch with world size 1 (the gpurun
    # boxes have one GPU): same code path as N>1, the all-reduce degenerates to a copy.
    use_dist = world > 1 or os.environ.get("QREC_FORCE_DIST") == "1"
    # Test hook for the 1-GPU development boxes: QREC_DIST_TEST_ONE_DEVICE=1 puts every rank on device 0 and uses the
    # gloo backend (RCCL refuses two ranks on one device), so that the N > 1 code path -- user sharding, delta
    # all-reduce, summed loss terms, per-rank device-side driver -- runs end to end on real hardware.  Numbers from
    # such a run are NOT bench results (the ranks share one GPU); the output line says so.
    one_device = os.environ.get("QREC_DIST_TEST_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    capi.init(local_rank)

    # ---- workload: resident in HBM before timing ------------------------------------------
    data = make_dataset(args.shape)
    U, I = data["n_users"], data["n_items"]
    indptr, items = to_csr(U, data["train_u"], data["train_i"])
    u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
    n = int(items.size)
    rng = np.random.default_rng(1000 + rank)
    P0 = (rng.random((U, DIM)) / 3).astype(np.float32)          # rand/3, iterativeRecommender.py:37-38
    Q0 = (np.random.default_rng(999).random((I, DIM)) / 3).astype(np.float32)  # same on all ranks
    tables = DeviceTables(P0, Q0, np.float32)
    sgd = BprSgd(tables, u, items, CSR(indptr, items), schedule=args.schedule)
    total = args.warmup + args.steps
    flush_every = args.flush_every or FLUSH_EVERY
    CHUNK = balanced_chunk(n)     # triplets per work item: the count that spreads evenly over the 4,096 persistent groups
    ev = [(capi.Event(), capi.Event()) for _ in range(total)]

    q_sync = stats_view = None
    if use_dist:   # replicated item table, reconciled once per step (qrec_amd/dist.py)
        from qrec_amd.dist import ReplicatedTableSync
        q_sync = ReplicatedTableSync(torch.as_tensor(tables.Q, device=torch.device("cuda", local_rank)))
        stats_view = torch.as_tensor(sgd.d_stats, device=torch.device("cuda", local_rank))   # [nll, sum P^2, sum Q^2] f64

    state = {"lr": LR0, "last": 0.0, "loss": 0.0}

    def between(stage: str):
        """N > 1, enqueue only: after the SGD kernel the ranks' Q deltas are summed (the path's one collective, RCCL
        all-reduce), after the local loss sums sum(-log sigma) and sum P*P are added over ranks (Q is replicated), so
        every rank's device-side driver takes the same decision."""
        if stage == "tables":
            q_sync.sync()
        else:
            dist.all_reduce(stats_view[0:2])

    def step(k: int):
        """sampler (side stream) | SGD kernel -> [N > 1: delta all-reduce] -> epoch close (BPR.py:40 loss terms,
        isConverged, updateLearningRate) all on the device; the host only enqueues.  tol = 0: the K timed steps all run."""
        sgd.take_prefetched_negatives(k)                            # BPR.py:35-37 (sampled under step k-1)
        sgd.epoch_device_async(REG_U, REG_I, MAX_LR, tol=0.0, chunk=CHUNK, variant=args.variant,
                               flush_every=flush_every, events=ev[k],   # BPR.py:45-53,40 + iterativeRecommender.py:88-104
                               between=between if use_dist else None)
        sgd.prefetch_negatives_device(2018, k + 1)                  # side stream, under the SGD kernel

    sgd.start_device_driver(LR0, log_capacity=total)

    def sync_all():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        capi.device_sync()

    sgd.prefetch_negatives_device(2018, 0)
    for k in range(args.warmup):
        step(k)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        step(k)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    drv = sgd.driver_state()
    if drv["failed"]:
        raise SystemExit("Loss = NaN or Infinity")            # iterativeRecommender.py:84-86
    assert drv["epochs"] == total and not drv["converged"], drv
    log = sgd.driver_log()
    state["loss"], state["lr"] = float(log[-1, 0]), drv["lr"]
    if os.environ.get("QREC_DIST_TEST_DUMP"):     # functional tests: every rank leaves its replica and its driver log behind
        np.savez(os.path.join(os.environ["QREC_DIST_TEST_DUMP"], f"rank{rank}.npz"), Q=tables.Q.numpy(), P=tables.P.numpy(),
                 log=log, lr=drv["lr"])
    kernel_ms = [ev[k][1].elapsed_ms_since(ev[k][0]) for k in range(args.warmup, total)]
    avg_kernel_ms = float(np.mean(kernel_ms))
    alg_bytes = n * bytes_per_triplet(DIM)

    if rank == 0:
        value = world * n * args.steps / elapsed
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("workload") == f"bpr-{args.shape}-d{DIM}-{args.schedule}":
                traffic = tj.get("bytes_per_launch")
        out = {
            **({"INVALID_AS_BENCH": "QREC_DIST_TEST_ONE_DEVICE: all ranks shared one GPU over gloo (functional test only)"} if one_device else {}),
            "metric": "BPR triplet-updates/sec", "value": value, "unit": "triplet-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BPR d={DIM} on synthetic Yelp2018-shape ({U}x{I}, {n} train triplets/epoch), "
                                   f"throughput mode (device Philox sampler + Hogwild atomic-delta SGD, {args.schedule}-major schedule)",
                       "triplets_per_step_per_gpu": n, "chunk": CHUNK,
                       "parallelism": "1 GPU" if world == 1 else f"user-sharded x{world}, replicated item table, per-step delta all-reduce (RCCL)",
                       "lr": LR0, "reg": REG_U, "final_loss": state["loss"], "final_lr": state["lr"],
                       "epoch_close": "device (no host sync inside the timed region)"},
            "roofline": {"bound": "hbm", "kernel": "bpr_hogwild_item_kernel<16,4>" if args.schedule == "item" else "bpr_hogwild_kernel<16,4,plain-load,atomic>",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_kernel_ms,
                         "note": "tables (17.8 MB) are L2/Infinity-Cache resident at this shape; the binding "
                                 "resource is the L2 atomic units (~1 dword/clk/channel), see DESIGN.md"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(u, items, indptr, I, U)
            out["vs_cpu_port"] = value / out["cpu_baseline"]["value"]
            # Recall@20 (the metric's second half): GPU throughput mode vs the order-exact CPU port on the
            # same negatives, same initial tables, same epochs and schedule; both ranked by the device ranker.
            Pg, Qg = tables.download(np.float32)
            Pc, Qc, loss_c = cpu_exact_order_reference(sgd, u, items, I, P0, Q0, total, 2018)
            r_gpu = evaluate_recall(Pg, Qg, data, indptr, items)
            r_cpu = evaluate_recall(Pc, Qc, data, indptr, items)
            out["recall_at_20"] = {"gpu_throughput_mode": r_gpu, "cpu_port_exact_order": r_cpu,
                                   "abs_diff": abs(r_gpu - r_cpu), "epochs": total,
                                   "final_loss_gpu": state["loss"], "final_loss_cpu": loss_c}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
