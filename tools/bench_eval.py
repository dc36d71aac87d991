#!/usr/bin/env python3
"""Full-rank evaluation at the Yelp2018 shape (31,668 users x 38,048 items, d=64, N=20, fp32): GPU time of one
qrec_score_topk call (HIP events around it), fused route and block route (QREC_EVAL_BLOCK_PATH=1 in a child process)."""
import json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run():
    from qrec_amd import capi
    from qrec_amd.capi import DeviceBuffer as DB
    from qrec_amd.interactions import CSR
    from qrec_amd.ranking import DeviceRanker
    from qrec_amd.synth import make_dataset, to_csr
    capi.init(0)
    d = make_dataset("yelp2018"); nu, ni = d["n_users"], d["n_items"]
    rng = np.random.default_rng(0)
    indptr, ind = to_csr(nu, d["train_u"], d["train_i"])
    U = (rng.random((nu, 64)) / 3 - 0.1).astype(np.float32); V = (rng.random((ni, 64)) / 3 - 0.1).astype(np.float32)
    rk = DeviceRanker(U, V, CSR(indptr, ind)); users = np.arange(nu, dtype=np.int32)
    rk.topk(users, 20)                                     # allocates scratch
    d_users = DB.from_numpy(users); e0, e1 = capi.Event(), capi.Event(); ts = []
    for _ in range(int(os.environ.get("REPS", "6"))):
        e0.record()
        capi.score_topk(rk.dU, rk.dV, rk.code, rk.d, rk.ld, ni, d_users, nu, rk.rated[0], rk.rated[1], 20, rk._scratch, rk._d_ids, rk._d_sc)
        e1.record(); e1.sync(); ts.append(e1.elapsed_ms_since(e0))
    t0 = time.perf_counter(); rk.topk(users, 20); wall = time.perf_counter() - t0
    return {"gpu_ms": float(np.median(ts)), "gpu_ms_all": ts, "scratch_GB": rk._scratch.nbytes / 1e9, "wall_ms_incl_readback": wall * 1e3,
            "gflop": 2 * nu * ni * 64 / 1e9, "tflops": 2 * nu * ni * 64 / np.median(ts) / 1e9}
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(json.dumps(run())); sys.exit(0)
    out = {"fused": run()}
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, QREC_EVAL_BLOCK_PATH="1"), capture_output=True, text=True)
    out["block"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-500:]
    print(json.dumps(out))
