#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes.  usage: summarize_pmc.py <out.json> COUNTER=<rocprof output dir> [COUNTER=<dir> ...]
One pass per counter (tools/gpu_steps.sh runs them separately, as /opt/skills/guides/MI355X_MICROARCH.md prescribes); FETCH_SIZE /
WRITE_SIZE are reported in the counters' own unit (KB; FETCH_SIZE reads half of the true bytes on gfx950 -- the x2 is applied where the
number is used, profiles/r01_fetch_size_calibration.txt)."""
import glob
import json
import sqlite3
import sys


def main():
    out_path, out = sys.argv[1], {}
    for spec in sys.argv[2:]:
        ctr, d = spec.split("=", 1)
        dbs = glob.glob(d + "/*_results.db") + glob.glob(d + "/*/*_results.db")
        if not dbs:
            print("no results.db under", d)
            continue
        con = sqlite3.connect(dbs[0])
        for name, n, avg, mx in con.execute("select kernel_name, count(*), avg(value), max(value) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
            k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
            e = out.setdefault(k, {})
            e[ctr + "_avg"], e[ctr + "_max"], e["dispatches"] = avg, mx, n
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -max(x for a, x in kv[1].items() if a.endswith("_avg"))):
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
