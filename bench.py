#!/usr/bin/env python3
"""bench.py -- BPR triplet-updates/sec on the MI355X hot path (BASELINE.json metric).

One EPOCH = one full pass of the reference's BPR epoch (model/ranking/BPR.py:29-43) over the synthetic
Yelp2018-shape interaction matrix (31,668 x 38,048, ~1.25 M train triplets, d=64): negative sampling for every
triplet (device Philox sampler) -> fused gather / dot / sigmoid / SGD scatter kernel (throughput mode) -> epoch-end
regulariser reductions -> loss, convergence test and bold-driver learning-rate update (the reference's isConverged)
on the device.  Inputs are resident in HBM before the timed region; the host only enqueues.

One STEP = `epochs_per_step` consecutive epochs: an epoch is 0.6 ms, so K = 20 single epochs would be a 12 ms sample;
the bench repeats epochs inside a step until the K timed steps cover >= --min-seconds (0.3 s).  `value` does not
depend on that grouping (triplets / second); `ms_per_step` is per step, `config.ms_per_epoch` per epoch.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # N > 1 typed like this: bench.py starts its own N ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>` with
the same arguments: one rank per GPU, rank 0's JSON line on the caller's stdout.  Under a launcher (WORLD_SIZE set) it is
one of the ranks.  The layout can be steered by flag or, for a caller whose command line is fixed, by environment:
--dist-mode / QREC_DIST_MODE (replicated | sharded), --scaling / QREC_SCALING (weak | strong).

N > 1, one process per GPU (qrec_amd/dist.py; collectives are RCCL bound directly by libqrec_hip.so, torch.distributed
/gloo is only the control plane).  Users (rows of P, triplets, sampler) are sharded by rank.  --dist-mode replicated
(default): item table replicated, ONE fused all-reduce of the per-rank deltas + loss terms per epoch.  --dist-mode
sharded: item table row-sharded, per batch an all-to-all of the distinct rows a rank's triplets touch and of their
updates.  --scaling weak (default): every rank its own 31,668 users; strong: the same 31,668 users split over ranks.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the SGD kernel's algorithmic bytes/launch over its mean launch time (HIP events on the launch
                  stream) against 8 TB/s HBM peak;  roofline_hbm_resident -- the same kernel on an HBM-sized problem
                  (a single-GPU slice of BASELINE config #4: 1.25 M x 1 M, d=128, 1.15 GB of tables);
  exact_mode   -- the order-exact mode (the one that meets the bit-exact-stream / 1e-5 numeric contract) timed on
                  the same workload;
  cpu_baseline -- the CPU port of the same epoch (oracle/, plain C, fp64, 1 thread) timed on this box's host cores
                  on a bounded sample; reference_loop_here: the reference's own interpreter-bound form (oracle/npref.py)
                  timed on this host; reference_python: the unmodified reference's figure (another host);
  recall_at_20 -- a fresh 25-epoch run of the timed mode against order-exact fp64 training on the same negatives;
  deferred_negatives -- the opt-in schedule `--schedule item-deferred` (one atomic row update per triplet) timed like the
                  main line, with its own Recall check (DESIGN.md s4 says why it is not the default);
  multi_gpu    -- N > 1: what RCCL reports per rank, kernel time per rank, collectives and bytes per epoch.
Sharded layout options: --shard-batch, --shard-pipeline, --plan-ahead, --no-plan-inside.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DIM = 64
LR0, MAX_LR, REG_U, REG_I = 0.01, 1.0, 0.001, 0.001   # config/BPR.conf:9-10
FLUSH_EVERY = 16
SEED = 2018


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
    from qrec_amd.interactions import CSR
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
    # ... and the interpreter-bound form the reference itself runs in (oracle/npref.py: one Python iteration and seven numpy row
    # statements per triplet, random.choice negatives), timed HERE on a bounded sample -- the reference cannot travel to this box
    from oracle import npref
    import random as _random
    Pn = rng.random((U, DIM)) / 3; Qn = rng.random((n_items, DIM)) / 3
    t1 = time.perf_counter()
    _, n_loop, _ = npref.bpr_epoch(Pn, Qn, indptr, i, n_items, LR0, REG_U, REG_I, rng=_random.Random(0), max_triplets=200_000)
    dt_loop = time.perf_counter() - t1
    return {"value": done / dt, "unit": "triplet-updates/s", "cores": 1, "kind": "port",
            "sample": f"{epochs} epochs ({done} triplets, {dt:.1f} s), C fp64 port of BPR.py:29-53 + CPython-stream sampler",
            "reference_loop_here": {"value": n_loop / dt_loop, "unit": "triplet-updates/s", "cores": 1, "kind": "port",
                                    "what": "the reference's own form -- per triplet one CPython iteration, random.choice, seven numpy row "
                                            "statements (BPR.py:28-53) -- restated in oracle/npref.py and timed on this host",
                                    "sample": f"{n_loop} triplets of the same epoch ({dt_loop:.1f} s), d = {DIM}, fp64"},
            "reference_python": {"value": 58930.0, "unit": "triplet-updates/s", "cores": 1,
                                 "host": "survey container (8 vCPU Xeon 2.1 GHz KVM), not this box: the Python reference cannot travel",
                                 "source": "BASELINE.md s2: unmodified BPR.trainModel, same shape, fp64"}}


def cpu_exact_order_reference(sgd, u, i, P0, Q0, epochs, seed):
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


def recall_check(capi, sgd, tables, data, u, items, indptr, P0, Q0, chunk, flush_every, variant, epochs=25):
    """The metric's second half: a fresh `epochs`-epoch throughput-mode run vs the order-exact CPU port on the same
    negatives, same initial tables and schedule; both ranked by the device ranker."""
    tables.upload(P0, Q0)
    sgd.start_device_driver(LR0, log_capacity=epochs)
    for k in range(epochs):
        sgd.sample_negatives_device(SEED, k)
        sgd.epoch_device_async(REG_U, REG_I, MAX_LR, tol=0.0, chunk=chunk, variant=variant, flush_every=flush_every)
    capi.device_sync()
    loss_g = float(sgd.driver_log()[-1, 0])
    Pg, Qg = tables.download(np.float32)
    Pc, Qc, loss_c = cpu_exact_order_reference(sgd, u, items, P0, Q0, epochs, SEED)
    r_gpu = evaluate_recall(Pg, Qg, data, indptr, items)
    r_cpu = evaluate_recall(Pc, Qc, data, indptr, items)
    return {"gpu_throughput_mode": r_gpu, "cpu_port_exact_order": r_cpu, "abs_diff": abs(r_gpu - r_cpu), "epochs": epochs,
            "final_loss_gpu": loss_g, "final_loss_cpu": loss_c}


def deferred_variant(capi, data, u, items, indptr, n_items, P0, Q0, chunk, flush_every, seed, main, runs=3, epochs=100):
    """The same workload under the opt-in schedule "item-deferred" (qrec_bpr_sgd_hogwild_item_major_deferred: one atomic row update
    per triplet instead of two, the negative-side terms applied by a second, j-ordered pass -- DESIGN.md s4), timed like the main
    line: `runs` fresh `epochs`-epoch trainings, sampler + j sort on the side stream, epoch close on the device, no host sync
    inside.  Reported NEXT to `value`, not as it: the deferral is a change of algorithm beyond Hogwild's (tests/test_gpu_bpr.py
    pins its effect on Recall@20: inside +-0.002 at BPR.conf's rate, outside at five times that), so BPR does not run it by default."""
    from qrec_amd.capi import DeviceBuffer
    from qrec_amd.engine import BprSgd, DeviceTables
    from qrec_amd.interactions import CSR
    t = DeviceTables(P0, Q0, np.float32)
    s = BprSgd(t, u, items, CSR(indptr, items), schedule="item-deferred", n_items=n_items, chunk=chunk)
    d_P0, d_Q0 = DeviceBuffer.from_numpy(t._pad(P0)), DeviceBuffer.from_numpy(t._pad(Q0))
    s.start_device_driver(LR0, log_capacity=epochs)
    d_drv0 = DeviceBuffer.from_numpy(s.d_drv.numpy())
    capi.device_sync()
    evs = [(capi.Event(), capi.Event()) for _ in range((runs + 1) * epochs)]
    k = 0
    s.prefetch_negatives_device(seed, 0)
    t0 = 0.0
    for r in range(runs + 1):                   # run 0: warm-up
        if r == 1:
            capi.device_sync(); t0 = time.perf_counter()
        capi.memcpy_d2d(t.P, d_P0, d_P0.nbytes, main); capi.memcpy_d2d(t.Q, d_Q0, d_Q0.nbytes, main)
        capi.memcpy_d2d(s.d_drv, d_drv0, d_drv0.nbytes, main)
        for _ in range(epochs):
            s.take_prefetched_negatives(k, main)
            s.epoch_device_async(REG_U, REG_I, MAX_LR, tol=0.0, chunk=chunk, stream=main, flush_every=flush_every, events=evs[k])
            s.prefetch_negatives_device(seed, k + 1)
            k += 1
    t_enq = time.perf_counter() - t0
    capi.device_sync()
    dt = time.perf_counter() - t0
    ms = float(np.mean([b.elapsed_ms_since(a) for a, b in evs[epochs:]]))
    alg = u.size * bytes_per_triplet(P0.shape[1])
    rec = recall_check(capi, s, t, data, u, items, indptr, P0, Q0, chunk, flush_every, capi.HW_DEFAULT)
    return {"schedule": "item-deferred", "value": u.size * runs * epochs / dt, "unit": "triplet-updates/s", "ms_per_epoch": dt / (runs * epochs) * 1e3,
            "kernels": "bpr_hogwild_item_kernel<16,4,defer> + bpr_deferred_negatives_kernel<16,4> (j order: rocPRIM radix sort on the sampler's stream)",
            "avg_launch_ms": ms, "roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "recall_at_20": rec,
            "host_enqueue_ms_per_epoch": t_enq / (runs * epochs) * 1e3,
            "default": False, "why_not_default": "Recall@20 vs exact-order training: within 0.002 at lr0 = 0.01, 0.0058 apart at lr0 = 0.05 (12-epoch paired runs, tests/test_gpu_bpr.py)"}


def exact_mode_rate(capi, u, items, indptr, n_items, P0, Q0):
    """The order-exact mode on the same workload: CPython-stream negatives from the native host replay, triplets
    applied strictly in the reference's order on the device (fp64 tables, the drop-in classes' default)."""
    import random
    from qrec_amd.engine import BprSgd, DeviceTables
    words = capi.state_from_python(random.Random(1).getstate())
    t0 = time.perf_counter()
    j = capi.mt_bpr_sample_epoch(words, indptr, items, n_items)
    t_sample = time.perf_counter() - t0
    t = DeviceTables(P0.astype(np.float64), Q0.astype(np.float64), np.float64)
    s = BprSgd(t, u, items); s.set_negatives(j)
    capi.device_sync()
    t0 = time.perf_counter()
    s.epoch_ordered(LR0, REG_U, REG_I)            # synchronous: schedule, upload, kernel, loss read back
    dt_single = time.perf_counter() - t0
    # steady state, as the BPR class runs it: the host side of epoch k + 1 (sampler replay, schedule, upload on a side stream)
    # under the kernel of epoch k
    side, n_ep = capi.Stream(), 4
    prep = s.prepare_ordered(j, slot=0, stream=side.handle); side.sync()
    capi.device_sync()
    t0 = time.perf_counter()
    for k in range(n_ep):
        s.run_prepared(prep, LR0, REG_U, REG_I)
        if k + 1 < n_ep:
            j = capi.mt_bpr_sample_epoch(words, indptr, items, n_items)
            prep = s.prepare_ordered(j, slot=(k + 1) & 1, stream=side.handle); side.sync()
        s.epoch_stats()                           # reads the loss terms back: the per-epoch host decision point
    dt = (time.perf_counter() - t0) / n_ep
    return {"value": u.size / dt, "unit": "triplet-updates/s", "dtype": "f64", "epoch_s": dt, "epochs_timed": n_ep,
            "epoch_s_unpipelined": dt_single, "host_sampler_s": t_sample,
            "parity": "index stream bit-exact vs the recorded reference run; P, Q 1e-10, loss 1e-11 (tests/test_gpu_bpr.py)"}


def hbm_resident_roofline(capi, schedule="user", sub_epochs=None):
    """BASELINE config #4, single-GPU slice (U=1.25 M, I=1 M, d=128, 25 M triplets, uniform items): 1.15 GB of tables,
    far beyond the 256 MiB Infinity Cache, so the gather+SGD kernel's traffic is real HBM traffic."""
    from qrec_amd.engine import BprSgd, DeviceTables
    rng = np.random.default_rng(0)
    U2, I2, n2, d2 = 1_250_000, 1_000_000, 25_000_000, 128
    u2 = np.sort(rng.integers(0, U2, n2, dtype=np.int32)); i2 = rng.integers(0, I2, n2, dtype=np.int32)
    blk = (rng.random((50_000, d2)) / 3).astype(np.float32)
    P2 = np.empty((U2, d2), np.float32); Q2 = np.empty((I2, d2), np.float32)
    for a in (P2, Q2):
        for k in range(0, a.shape[0], 50_000):
            a[k:k + 50_000] = blk[:min(50_000, a.shape[0] - k)]
    t = DeviceTables(P2, Q2, np.float32); s = BprSgd(t, u2, i2, None, schedule=schedule, sub_epochs=sub_epochs)
    s.set_negatives(rng.integers(0, I2, n2, dtype=np.int32))
    e0, e1 = capi.Event(), capi.Event(); ts = []
    for _ in range(5):
        e0.record(); s.epoch_throughput_async(LR0, REG_U, REG_I); e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
    ms = float(np.median(ts[1:])); alg = n2 * bytes_per_triplet(d2)
    return {"workload": f"BPR d={d2}, {U2}x{I2}, {n2} triplets/epoch, {schedule}-major" + (f", {sub_epochs} sub-epochs" if sub_epochs else "")
                        + " (config #4 single-GPU slice, tables 1.15 GB)",
            "bound": "hbm", "achieved": alg / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / ms / 1e6 / HBM_PEAK_GBPS,
            "avg_launch_ms": ms, "algorithmic_bytes_per_launch": alg, "triplet_updates_per_s": n2 / ms * 1e3}


def launch_own_ranks(n: int):
    """`python bench.py --gpus N` outside a launcher: become `torch.distributed.run` with N ranks of this very command line
    (exec, so rank 0's single JSON line is the only thing on the caller's stdout and the exit code is the launcher's)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                   # the launcher would set it anyway, with a warning on stderr
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="yelp2018")
    ap.add_argument("--min-seconds", type=float, default=0.3, help="epochs are repeated inside a step until the timed region lasts this long")
    ap.add_argument("--epochs-per-step", type=int, default=0, help="fix the inner repeat instead of calibrating it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip exact_mode / roofline_hbm_resident / recall_at_20")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flush-every", type=int, default=0, help="item-major schedule: triplets between flushes of the register-resident Q[i] (0 = default)")
    ap.add_argument("--schedule", choices=("item", "user", "item-deferred"), default=os.environ.get("QREC_BENCH_SCHEDULE", "item"),
                    help="visiting order of the epoch's triplets in the Hogwild kernel (DESIGN.md s4)")
    ap.add_argument("--dist-mode", choices=("replicated", "sharded"), default=os.environ.get("QREC_DIST_MODE", "replicated"))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("QREC_SCALING", "weak"))
    ap.add_argument("--shard-batch", type=int, default=1 << 20,
                    help="sharded mode: triplets per exchange batch and rank (2^20: at most 2^21 distinct item rows in a rank's cache, 1 GiB at "
                         "d = 128; an epoch of 2^19 triplets or more is split into at least two batches, so that the next epoch's plan hides "
                         "in front of the last one)")
    ap.add_argument("--shard-pipeline", action="store_true", default=os.environ.get("QREC_SHARD_PIPELINE") == "1",
                    help="sharded mode: fetch batch k + 1 under batch k's SGD kernel, on a second stream and communicator (one more batch of "
                         "staleness).  Off by default: at the Yelp2018 shape the fetch is 10 MB per batch and the gather/copy kernels "
                         "running beside the atomic-bound SGD grid cost it more than they hide (world 1: 0.85 vs 0.75 ms/epoch, "
                         "profiles/r03_sharded_world1.json); it is for shapes whose exchange outlasts the batch's SGD (config #4)")
    ap.add_argument("--no-shard-pipeline", action="store_true", help="(accepted for compatibility: the default)")
    ap.add_argument("--no-plan-inside", action="store_true",
                    help="sharded mode: plan every epoch at its own start, with the host waiting for the stream to run dry (default: the "
                         "next epoch's plan is enqueued in front of the current epoch's last batch, its row counts read back behind an event)")
    ap.add_argument("--plan-ahead", action="store_true", default=os.environ.get("QREC_SHARD_PLAN_AHEAD") == "1",
                    help="sharded mode: plan every epoch (distinct rows per owner, one host read-back, id exchange) under the PREVIOUS epoch, on "
                         "a third stream and communicator (default: at its own start, on the training stream -- measured at world 1: the plan "
                         "kernels running beside the atomic-bound SGD grid cost it more than the host round trip they hide, "
                         "profiles/r03_sharded_world1.json)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_own_ranks(args.gpus)           # does not return

    # stdout carries ONE line, the result.  Libraries print there too -- RCCL writes its version banner to C stdio's
    # stdout, flushed at exit, i.e. AFTER a Python print -- so file descriptor 1 is pointed at stderr for the whole run and
    # the JSON line goes to the saved original descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # QREC_FORCE_DIST=1 drives the multi-GPU branch with world size 1 (the gpurun boxes have one GPU): same code path
    # as N > 1 -- delta / apply kernels, the real RCCL communicator -- the collectives degenerate to copies.
    use_dist = world > 1 or os.environ.get("QREC_FORCE_DIST") == "1"
    # Test hook for the 1-GPU development boxes: QREC_DIST_TEST_ONE_DEVICE=1 puts every rank on device 0 with the staged
    # gloo transport (RCCL refuses two ranks on one device).  Not a bench result; the output line says so.
    one_device = os.environ.get("QREC_DIST_TEST_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    control = comm = qd = None
    from qrec_amd import capi
    capi.load()       # BEFORE torch is imported: libqrec_hip.so then binds /opt/rocm's HIP runtime (and, next to it, its librccl)
                      # and not the older one bundled with torch, on which the sampler's side stream does not overlap the
                      # SGD kernel (measured: 0.67 vs 0.61 ms per epoch).  torch is the CPU control plane only.
    if use_dist:
        from qrec_amd import dist as qd
        control = qd.ControlPlane.from_env()
    from qrec_amd.capi import DeviceBuffer
    from qrec_amd.engine import BprSgd, DeviceTables, balanced_chunk
    from qrec_amd.interactions import CSR
    from qrec_amd.synth import make_dataset, to_csr
    capi.init(local_rank)
    if use_dist and os.environ.get("QREC_BENCH_NO_COMM") == "1":      # diagnosis only: the multi-GPU code path without a communicator
        comm = type("NoComm", (), {"world": 1, "rank": 0, "allreduce": lambda *a, **k: None, "allreduce_pair": lambda *a, **k: None,
                                   "destroy": lambda self: None})()
    elif use_dist:
        comm = qd.make_comm(control)

    # ---- workload: resident in HBM before timing ------------------------------------------
    data = make_dataset(args.shape)
    U, I = data["n_users"], data["n_items"]
    indptr, items = to_csr(U, data["train_u"], data["train_i"])
    u = np.repeat(np.arange(U, dtype=np.int32), np.diff(indptr)).astype(np.int32)
    n_full = int(items.size)
    Q0 = (np.random.default_rng(999).random((I, DIM)) / 3).astype(np.float32)    # same on all ranks
    strong = args.scaling == "strong" and world > 1
    if strong:     # the SAME users split over the ranks: rank r trains the r-th contiguous block (qrec_amd/dist.py)
        lo, hi, l_indptr, l_items = qd.shard_positive_csr(indptr, items, world, rank)
        l_u = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(l_indptr)).astype(np.int32)
        P0 = (np.random.default_rng(1000).random((U, DIM)) / 3).astype(np.float32)[lo:hi]
    else:          # weak: every rank its own population of U users with the same interaction structure
        l_indptr, l_items, l_u = indptr, items, u
        P0 = (np.random.default_rng(1000 + rank).random((U, DIM)) / 3).astype(np.float32)   # rand/3, iterativeRecommender.py:37-38
    n = int(l_items.size)
    sharded = use_dist and args.dist_mode == "sharded"
    Q0_local = qd.shard_item_rows(Q0, world, rank) if sharded else Q0
    tables = DeviceTables(P0, Q0_local, np.float32)
    flush_every = args.flush_every or FLUSH_EVERY
    CHUNK = balanced_chunk(n)     # triplets per work item: the count that spreads evenly over the 4,096 persistent groups
    n_batches = qd.agree_on_batches(control, n, args.shard_batch, split_from=1 << 19) if sharded else 1
    sgd = BprSgd(tables, l_u, l_items, CSR(l_indptr, l_items), schedule=args.schedule, n_items=I, batches=n_batches, chunk=CHUNK)
    sampler_seed = SEED + 7919 * rank

    dstep = None
    if use_dist and sharded:
        pipe = None
        if args.shard_pipeline and not args.no_shard_pipeline and os.environ.get("QREC_BENCH_NO_COMM") != "1":
            # a second communicator for the fetch stream: two collectives of ONE communicator must not be in flight on two streams
            pipe = (comm if one_device else qd.make_comm(control), capi.Stream())
        ahead = None
        if args.plan_ahead and os.environ.get("QREC_BENCH_NO_COMM") != "1":
            # ... and a third one for the plan of the NEXT epoch (row counts, id exchange), which runs under the current epoch
            ahead = (comm if one_device else qd.make_comm(control), capi.Stream())
        dstep = qd.ShardedStep(comm, qd.ShardedItemExchange(comm, I, tables.ld, tables.Q, pipeline=pipe, plan_ahead=ahead), n_batches,
                               plan_inside=not args.no_plan_inside)
    elif use_dist:
        dstep = qd.ReplicatedStep(comm, qd.ReplicatedTableSync(comm, tables.Q))
    # device copies of the initial state: every step restarts training from it (see step())
    d_P0, d_Q0 = DeviceBuffer.from_numpy(tables._pad(P0)), DeviceBuffer.from_numpy(tables._pad(Q0_local))

    ev, pool = [], []
    counter = {"epoch": 0}
    # The step's kernels and collectives run on an explicit non-blocking stream, not the null stream: the legacy null
    # stream synchronises implicitly with every blocking stream of the process, and an RCCL communicator brings its own --
    # measured at world 1 (QREC_FORCE_DIST=1): 0.670 ms/epoch on the null stream, 0.626 on this one, 0.610 with no
    # communicator in the process at all (QREC_BENCH_NULL_STREAM=1 / QREC_BENCH_NO_COMM=1 reproduce the two ends).
    main = None if os.environ.get("QREC_BENCH_NULL_STREAM") == "1" else capi.Stream()

    def epoch():
        """sampler (side stream) | SGD kernel -> [N > 1: collectives] -> epoch close (BPR.py:40 loss terms, isConverged,
        updateLearningRate) all on the device; the host only enqueues (sharded mode: plus ONE read-back of the exchange's
        row counts per epoch).  tol = 0: every epoch runs."""
        k = counter["epoch"]; counter["epoch"] += 1
        pair = pool.pop() if pool else (capi.Event(), capi.Event())
        ev.append(pair)
        sgd.take_prefetched_negatives(k, main)                      # BPR.py:35-37 (sampled under epoch k-1)
        if sharded:
            dstep.prepare(sgd, main)
        if sharded:      # the next epoch's sampler is enqueued from inside (same place on the device: behind the start event) -- the
            # epoch's last batch is preceded by the next epoch's plan, which waits for those negatives
            sgd.epoch_device_async(REG_U, REG_I, MAX_LR, tol=0.0, chunk=CHUNK, variant=args.variant, stream=main, flush_every=flush_every,
                                   events=pair, dist=dstep, after_start=lambda: sgd.prefetch_negatives_device(sampler_seed, k + 1))
            dstep.prepare_ahead(sgd)                                # --plan-ahead: on the plan stream instead
            return
        sgd.epoch_device_async(REG_U, REG_I, MAX_LR, tol=0.0, chunk=CHUNK, variant=args.variant, stream=main,
                               flush_every=flush_every, events=pair, dist=dstep)   # BPR.py:45-53,40 + iterativeRecommender.py:88-104
        sgd.prefetch_negatives_device(sampler_seed, k + 1)          # side stream, under the SGD kernel

    def sync_all():
        if use_dist:
            control.barrier()
        capi.device_sync()          # hipDeviceSynchronize on the runtime all of this process's GPU work runs on
        if use_dist:
            control.barrier()

    def restart():
        """Every step is a fresh training run of `inner` epochs from the initial tables and learning rate (device-to-
        device copies inside the timed region -- extra work, not skipped work): the bold driver (BPR.conf: -max 1) halves
        the rate whenever sampling noise raises the loss, so a single run of many hundred epochs ends at a vanishing
        rate, while the reference trains 100 epochs at most; this keeps every timed epoch in the regime of a real run."""
        capi.memcpy_d2d(tables.P, d_P0, d_P0.nbytes, main); capi.memcpy_d2d(tables.Q, d_Q0, d_Q0.nbytes, main)
        if dstep is not None and dstep.mode == "replicated":
            capi.memcpy_d2d(dstep.sync_q.start, d_Q0, d_Q0.nbytes, main)
        capi.memcpy_d2d(sgd.d_drv, d_drv0, d_drv0.nbytes, main)

    # calibration: how many epochs make a step, so that K steps last >= --min-seconds (same on every rank)
    cal = 5
    inner_max = 400
    sgd.start_device_driver(LR0, log_capacity=args.epochs_per_step or inner_max)
    d_drv0 = DeviceBuffer.from_numpy(sgd.d_drv.numpy())
    capi.device_sync()            # set-up work sits on the null stream; the epochs run on `main`
    sgd.prefetch_negatives_device(sampler_seed, 0)
    epoch(); epoch(); sync_all()
    t0 = time.perf_counter()
    for _ in range(cal - 2):
        epoch()
    sync_all()
    t_epoch = (time.perf_counter() - t0) / (cal - 2)
    if args.epochs_per_step:
        inner = args.epochs_per_step
    else:
        inner = min(inner_max, max(1, math.ceil(args.min_seconds / (args.steps * t_epoch))))
        if use_dist:
            inner = int(control.allreduce_host(np.array([inner], dtype=np.int64), op="max")[0])

    def step():
        restart()
        for _ in range(inner):
            epoch()

    pool.extend((capi.Event(), capi.Event()) for _ in range((args.warmup + args.steps) * inner))   # not inside the timed loop
    for _ in range(args.warmup):
        step()
    sync_all()
    first_timed = counter["epoch"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        elapsed = float(control.allreduce_host(np.array([elapsed]), op="max")[0])
    total = counter["epoch"]

    drv = sgd.driver_state()
    if drv["failed"]:
        raise SystemExit("Loss = NaN or Infinity")            # iterativeRecommender.py:84-86
    assert drv["epochs"] == inner and not drv["converged"], drv
    log = sgd.driver_log()
    final_loss, final_lr = float(log[-1, 0]), drv["lr"]
    if os.environ.get("QREC_DIST_TEST_DUMP"):     # functional tests: every rank leaves its tables and its driver log behind
        np.savez(os.path.join(os.environ["QREC_DIST_TEST_DUMP"], f"rank{rank}.npz"), Q=tables.Q.numpy(), P=tables.P.numpy(),
                 log=log, lr=drv["lr"], epochs_per_step=inner)
    kernel_ms = [ev[k][1].elapsed_ms_since(ev[k][0]) for k in range(first_timed, total)]
    avg_kernel_ms = float(np.mean(kernel_ms))
    alg_bytes = n * bytes_per_triplet(DIM)
    moved = None
    if sharded:
        moved = float(control.allreduce_host(np.array([dstep.exchange.bytes_moved / max(total, 1)]))[0])
    multi = None
    if use_dist:
        # what the run looked like from every rank: the SGD kernel's mean launch time per rank (HIP events on the rank's own
        # stream) and what RCCL itself reports for the communicator (ncclCommCount / UserRank / CuDevice)
        q = comm.query() if hasattr(comm, "query") else {"ranks": comm.world, "rank": comm.rank, "device": local_rank}
        per_rank = control.allgather_host(np.array([avg_kernel_ms, float(q["ranks"]), float(q["rank"]), float(q["device"]), float(n)]))
        lib = None
        if hasattr(comm, "query"):
            path, ver = capi.comm_library()
            lib = {"library": path, "version": ver}
        q_floats = int(tables.Q.nbytes // 4)
        payload = None if sharded else q_floats * 4 + 24          # the fused all-reduce: item-table deltas + 3 fp64 loss terms
        multi = {"rccl_ranks": int(per_rank[:, 1].min()), "rccl_ranks_agree": bool((per_rank[:, 1] == per_rank[0, 1]).all()),
                 "rank_devices": [int(x) for x in per_rank[:, 3]], "rccl": lib,
                 "transport": "rccl" if hasattr(comm, "query") else type(comm).__name__,
                 "kernel_ms_per_rank": {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max()),
                                        "all": [float(x) for x in per_rank[:, 0]]},
                 "triplets_per_epoch_per_rank": [int(x) for x in per_rank[:, 4]],
                 "collectives_per_epoch": ({"all_to_all_batches": dstep.n_batches, "calls": 3 * dstep.n_batches + 1,
                                            "bytes_leaving_all_ranks": moved} if sharded else
                                           {"all_reduce": 1, "payload_bytes_per_rank": payload,
                                            "ring_wire_bytes_per_rank": (2.0 * (world - 1) / world * payload) if world > 1 else 0.0})}

    if rank == 0:
        n_job = n_full if strong else world * n
        value = n_job * args.steps * inner / elapsed
        achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile) and not sharded:
            tj = json.load(open(tfile))
            if tj.get("workload") == f"bpr-{args.shape}-d{DIM}-{args.schedule}":
                traffic = tj.get("bytes_per_launch")
        kernel = {"item": "bpr_hogwild_item_kernel<16,4>", "user": "bpr_hogwild_kernel<16,4,plain-load,atomic>",
                  "item-deferred": "bpr_hogwild_item_kernel<16,4,defer> + bpr_deferred_negatives_kernel<16,4>"}[args.schedule]
        if world == 1:
            par = "1 GPU" + (f" (QREC_FORCE_DIST: {args.dist_mode} multi-GPU path at world 1)" if use_dist else "")
        elif sharded:
            par = f"users x{world}, item table row-sharded x{world}: per-batch RCCL all-to-all of distinct rows + their updates"
        else:
            par = f"users x{world}, item table replicated: one fused RCCL all-reduce of deltas + loss terms per epoch"
        out = {
            **({"INVALID_AS_BENCH": "QREC_DIST_TEST_ONE_DEVICE: all ranks shared one GPU over gloo (functional test only)"} if one_device else {}),
            "metric": "BPR triplet-updates/sec", "value": value, "unit": "triplet-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BPR d={DIM} Yelp2018-shape {U}x{I}" if args.shape == "yelp2018" else f"BPR d={DIM} {args.shape} {U}x{I}",
                       "mode": f"throughput: device Philox sampler + Hogwild atomic-delta SGD, {args.schedule}-major",
                       "triplets_per_epoch_per_gpu": n, "epochs_per_step": inner,
                       "step": f"a fresh {inner}-epoch training run from the initial tables and learning rate", "ms_per_epoch": elapsed / (args.steps * inner) * 1e3,
                       "timed_seconds": elapsed, "chunk": CHUNK, "parallelism": par,
                       "lr": LR0, "reg": REG_U, "final_loss": final_loss, "final_lr": final_lr,
                       "epoch_close": "device (no host sync inside the timed region)" if not sharded else "device; one row-count read-back per epoch for the exchange",
                       "dist_mode": args.dist_mode if use_dist else None,
                       **({"xgmi_bytes_per_epoch_all_ranks": moved, "batches_per_epoch": dstep.n_batches,
                           "fetch_pipelined": dstep.exchange.pipeline is not None,
                           "plan": ("ahead, on a plan stream" if dstep.exchange.plan_ahead is not None else
                                    "inside the previous epoch, in front of its last batch" if dstep.plan_inside and dstep.n_batches >= 2 else
                                    "at the epoch's start")} if sharded else {})},
            **({"multi_gpu": multi} if multi is not None else {}),
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": ("profiles/hbm_traffic.json: rocprofv3 PMC passes of this command run by the builder (static; "
                                            "not re-measured in this run)") if traffic is not None else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_kernel_ms,
                         "note": ("events bracket the epoch's batches incl. their exchanges" if sharded else
                                  "tables (17.8 MB) are cache resident at this shape; bound = L2 atomic units, see DESIGN.md"),
                         **({"atomic_unit_floor": {"ms": 0.547, "of_this_kernel": 0.547 / avg_kernel_ms, "source": "profiles/r03_ubench_atomics4.txt "
                                                   "(static, builder-measured): this epoch's two atomic row updates per triplet ALONE, no loads, no arithmetic "
                                                   "= 308 G dword atomics/s = one dword per clock on each of the 128 L2 channels"}}
                            if args.schedule == "item" and args.shape == "yelp2018" and not use_dist else {})},
        }
        if world == 1 and not use_dist:
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(u, items, indptr, I, U)
                out["vs_cpu_port"] = value / out["cpu_baseline"]["value"]
                out["vs_reference_loop_here"] = value / out["cpu_baseline"]["reference_loop_here"]["value"]
            if not args.no_extras:
                out["recall_at_20"] = recall_check(capi, sgd, tables, data, u, items, indptr, P0, Q0, CHUNK, flush_every, args.variant)
                out["exact_mode"] = exact_mode_rate(capi, u, items, indptr, I, P0, Q0)
                # the main run's objects go first, their streams with them: a process maps its streams onto a handful of hardware
                # queues (4 by default), and with the main run's two still alive the leg's sampler stream shared a queue with its
                # training stream -- sampler and sort ran BEHIND the SGD kernels instead of under them (0.57 instead of 0.49 ms/epoch)
                del sgd, tables, d_P0, d_Q0
                if args.schedule == "item" and args.shape == "yelp2018":
                    out["deferred_negatives"] = deferred_variant(capi, data, u, items, indptr, I, P0, Q0, CHUNK, flush_every, sampler_seed, main)
                out["roofline_hbm_resident"] = hbm_resident_roofline(capi)
                if "deferred_negatives" in out:     # the opt-in schedule on the HBM-resident slice, in four sub-epochs (free at this size)
                    out["deferred_negatives"]["roofline_hbm_resident"] = hbm_resident_roofline(capi, schedule="item-deferred", sub_epochs=4)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        control.barrier()
        for extra in ((dstep.exchange.pipeline, dstep.exchange.plan_ahead) if sharded else ()):
            if extra is not None and extra[0] is not comm:
                extra[0].destroy()
        if comm is not None:
            comm.destroy()
        control.shutdown()


if __name__ == "__main__":
    main()
