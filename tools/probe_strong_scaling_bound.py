#!/usr/bin/env python3
"""What ONE rank of an N-rank strong-scaling run of the Yelp2018-shape BPR epoch costs on its own GPU, links excluded: rank 0's block of
users, N reconciliations per epoch (the default, dist.reconciliations_per_epoch) through an identity "communicator" -- i.e. the SGD batches,
the delta / apply kernels and the epoch close, but no bytes over xGMI.  t(1) / t(N) is the ceiling of the strong-scaling speed-up before any
link time; the collectives' wire bytes are printed next to it with the arithmetic of bench.py (NOT a measurement of the links: the gpurun
boxes have one GPU).  One JSON object."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qrec_amd import capi, dist as qd
from tools import paired_recall as PR


class NoLinks:
    """capi.Comm's interface with a world of `world` that moves nothing: every collective is the identity on this rank's data"""
    def __init__(self, world): self.world, self.rank = world, 0
    def allreduce(self, *a, **k): pass
    def allreduce_pair(self, *a, **k): pass


capi.init(0)
shape = sys.argv[1] if len(sys.argv) > 1 else "yelp2018"
d = PR.load_dataset(shape)
P0, Q0 = PR.initial_tables(d, 3)
out = {"shape": shape, "users": d["n_users"], "items": d["n_items"], "triplets_per_epoch": int(d["items"].size), "ranks": {}}
# plan: JSON list of {"N":, "K":, "groups": grid of a batch launch, "lo": min chunk, "rounds": engine.grid_for_epoch's minimum}
PLAN = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else json.loads(os.environ["PROBE_PLAN"]) if os.environ.get("PROBE_PLAN") else \
    [{"N": N, "K": K} for N in (1, 2, 4, 8) for K in sorted({1, 2, N})]
import qrec_amd.engine as E
_launch_chunk = E.launch_chunk
for cfg in PLAN:
        N, K, bg, lo = cfg["N"], cfg["K"], cfg.get("groups", 4096), cfg.get("lo", 4)
        E.launch_chunk = lambda n, chunk, groups=4096, lo_=lo, **kw: _launch_chunk(n, chunk, groups=groups, lo=lo_)
        t, sgd, chunk, lo_u, hi_u, _groups = PR.build_rank(d, "item", N, 0, "replicated", P0, Q0, syncs=K, rounds=cfg.get("rounds", 0))
        sgd.batch_groups = bg
        step = qd.ReplicatedStep(NoLinks(N), qd.ReplicatedTableSync(NoLinks(N), t.Q)) if N > 1 else None
        stream = capi.Stream()
        epochs = 60
        sgd.start_device_driver(0.01, log_capacity=epochs + 10)
        capi.device_sync()
        def run(n0, n1):
            for k in range(n0, n1):
                sgd.sample_negatives_device(7, k, stream)
                sgd.epoch_device_async(PR.REG, PR.REG, PR.MAX_LR, tol=0.0, chunk=chunk, flush_every=PR.FLUSH, stream=stream, dist=step, groups=_groups if K == 1 else 0)
        run(0, 10); stream.sync()
        t0 = time.perf_counter(); run(10, 10 + epochs - 10); stream.sync(); dt = (time.perf_counter() - t0) / (epochs - 10)
        payload = d["n_items"] * t.ld * 4 + 24
        inner = payload
        wire = 2.0 * (N - 1) / N * (payload + (K - 1) * inner) if N > 1 else 0.0
        key = f"N={N},K={K}" + (f",groups={bg}" if bg != 4096 else "") + (f",lo={lo}" if lo != 4 else "") + (f",rounds={cfg['rounds']}" if cfg.get("rounds") else "")
        out["ranks"][key] = {"rank0_triplets": int(sgd.n), "reconciliations_per_epoch": K, "ms_per_epoch_no_links": dt * 1e3,
                                        "ring_wire_MB_per_rank_per_epoch": wire / 1e6, "collectives_per_epoch": K,
                                        "link_ms_arithmetic": {"one_ring_153GBps": wire / 153e9 * 1e3, "seven_rings_1071GBps": wire / 1071e9 * 1e3}}
        del sgd, t, step
base = out["ranks"]["N=1,K=1"]["ms_per_epoch_no_links"]
for k, v in out["ranks"].items():
    v["speedup_ceiling_no_links"] = base / v["ms_per_epoch_no_links"]
    v["speedup_with_link_arithmetic"] = {a: base / (v["ms_per_epoch_no_links"] + b) for a, b in v["link_ms_arithmetic"].items()}
out["_note"] = ("the epoch here includes the (serial, same-stream) device sampler, unlike bench.py's pipelined epoch: ratios are what matter.  "
                "Link times are arithmetic (bytes / nominal link rate), latency per collective not included")
print(json.dumps(out))
for k, v in out["ranks"].items():
    print(k, "ms %.4f" % v["ms_per_epoch_no_links"], "speed-up ceiling %.2f" % v["speedup_ceiling_no_links"], file=sys.stderr)
