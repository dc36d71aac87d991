#!/usr/bin/env python3
"""Turn rocprofv3 (rocpd sqlite) outputs under gpurun_out/ into the small text/JSON summaries
committed under profiles/.  usage: tools/summarize_prof.py <round-tag> [workload-tag]"""
import json, os, sqlite3, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "bpr-yelp2018-d64-item"
G = "gpurun_out"; os.makedirs("profiles", exist_ok=True)
def q(db, sql):
    con = sqlite3.connect(db); rows = list(con.execute(sql)); con.close(); return rows
lines = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline   [{tag}, {workload}]",
         "# columns: calls  total_us  avg_us  pct  kernel"]
stats = q(f"{G}/prof_stats/{tag}_results.db", "select name,total_calls,total_duration,average,percentage from top_kernels")
for name, calls, tot, avg, pct in stats:
    lines.append(f"{calls:6d} {tot:12.1f} {avg:10.3f} {pct:6.2f}  {name[:150]}")
res = q(f"{G}/prof_stats/{tag}_results.db", "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels group by name")
lines.append(""); lines.append("# resources: vgpr agpr sgpr lds grid_x wg_x kernel")
for r in res:
    lines.append(f"{r[1]} {r[2]} {r[3]} {r[4]} {r[5]} {r[6]}  {r[0][:120]}")
open(f"profiles/{tag}_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
out = {"workload": workload, "round": tag, "counters": {}}
for sub, ctr in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    db = f"{G}/{sub}/{tag}_results.db"
    if not os.path.exists(db): continue
    for name, n, avg in q(db, f"select kernel_name, count(*), avg(value) from counters_collection where counter_name='{ctr}' group by kernel_name"):
        out["counters"].setdefault(name[:160], {})[ctr + "_KB_avg"] = avg
        out["counters"][name[:160]]["dispatches_" + ctr] = n
hot = [k for k in out["counters"] if "bpr_hogwild" in k]
if hot:
    c = out["counters"][hot[0]]
    # MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports 1/2 of the bytes read (verified for
    # this kernel's 64-B-segment dword pattern with tools/ubench/fetch_calib.hip: 1.000 GiB
    # reported for 2.000 GiB read); WRITE_SIZE is taken as is (it equals the atomic payload).
    out["kernel"] = hot[0]
    out["bytes_per_launch"] = (2 * c["FETCH_SIZE_KB_avg"] + c["WRITE_SIZE_KB_avg"]) * 1024
    out["fetch_bytes_corrected"] = 2 * c["FETCH_SIZE_KB_avg"] * 1024
    out["write_bytes"] = c["WRITE_SIZE_KB_avg"] * 1024
    out["method"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (with --kernel-trace only); bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB"
json.dump(out, open(f"profiles/{tag}_hbm_counters.json", "w"), indent=1)
json.dump({k: out[k] for k in ("workload", "round", "kernel", "bytes_per_launch", "fetch_bytes_corrected", "write_bytes", "method") if k in out},
          open("profiles/hbm_traffic.json", "w"), indent=1)
print(open(f"profiles/{tag}_kernel_stats.txt").read()); print(json.dumps(out, indent=1)[:1500])
