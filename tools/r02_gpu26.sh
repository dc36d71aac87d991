#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_sharded
QREC_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d $O/prof_sharded -o sh -- python $R/bench.py --no-cpu-baseline --no-extras --dist-mode sharded --steps 10 --warmup 2 > $O/prof_sharded.log 2>&1; echo "exit $?"
python - <<'P'
import sqlite3, json
d=json.loads([l for l in open('/root/repo/gpurun_out/prof_sharded.log') if l.startswith('{"metric')][-1]); print(d["config"]["ms_per_epoch"], d["config"].get("batches_per_epoch"), d["config"].get("epochs_per_step"))
con=sqlite3.connect('/root/repo/gpurun_out/prof_sharded/sh_results.db')
rows=list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
for name,calls,t,avg,pct in rows[:16]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:80]}")
P
