#!/usr/bin/env python3
"""Bytes past the XCD L2s per launch of the forward SpMM (layer sum fused in) from two rocprofv3 --pmc passes over tools/bench_lightgcn.py
(FETCH_SIZE and WRITE_SIZE, separate runs): the tool's own timing loop is the LAST 20 dispatches of spmm_kernel, one launch type.
usage: summarize_spmm_counters.py <out.json> <shape>=<fetch dir>,<write dir> [<shape>=...]
FETCH_SIZE reads half of the true bytes of 16 B/lane reads on gfx950 (MI355X_MICROARCH.md; profiles/r01_fetch_size_calibration.txt): doubled here."""
import glob
import json
import sqlite3
import sys


def last20(d, ctr):
    db = (glob.glob(d + "/*_results.db") + glob.glob(d + "/*/*_results.db"))[0]
    con = sqlite3.connect(db)
    rows = [v for (v,) in con.execute("select value from counters_collection where counter_name=? and kernel_name like '%spmm_kernel%' order by dispatch_id", (ctr,))]
    return sum(rows[-20:]) / 20.0, len(rows), sum(rows) / max(len(rows), 1)


out = {"_what": __doc__.split("usage")[0].strip()}
for spec in sys.argv[2:]:
    shape, dirs = spec.split("=", 1)
    fdir, wdir = dirs.split(",")
    f20, n, fall = last20(fdir, "FETCH_SIZE"); w20, _, wall = last20(wdir, "WRITE_SIZE")
    mb = (2 * f20 + w20) * 1024 / 1e6
    alg = None
    for line in open(fdir + ".log"):
        if line.startswith("{"):
            j = json.loads(line)
            alg = 8 * int(j["workload"].split("nnz=")[1]) + 4 * (int(j["workload"].split("N=")[1].split()[0]) + 1) + 2 * int(j["workload"].split("N=")[1].split()[0]) * 64 * 4
            spmm_ms = j["spmm_ms"]
    out[shape] = {"spmm_forward_with_layer_sum": {"FETCH_SIZE_KB_avg": f20, "WRITE_SIZE_KB_avg": w20, "l2_miss_MB_per_launch": mb, "dispatches_seen": n,
                                                  "algorithmic_MB": alg / 1e6 if alg else None, "over_fetch_vs_algorithmic": mb * 1e6 / alg if alg else None,
                                                  "spmm_ms_in_the_fetch_pass": spmm_ms},
                  "spmm_in_step_average": {"l2_miss_MB_per_launch": (2 * fall + wall) * 1024 / 1e6}}
    print(shape, json.dumps(out[shape]["spmm_forward_with_layer_sum"]))
json.dump(out, open(sys.argv[1], "w"), indent=1)
