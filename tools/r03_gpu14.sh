#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_shard
QREC_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_shard -o shard -- python $R/bench.py --dist-mode sharded --no-shard-pipeline --steps 3 --warmup 1 --epochs-per-step 20 --no-cpu-baseline --no-extras > $O/prof_shard.log 2>&1; echo "exit $?"
tail -2 $O/prof_shard.log | cut -c1-400
cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/prof_shard/*results.db")[0]
con = sqlite3.connect(db)
print([r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")][:80])
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"): print(r[1], round(r[2]/1e3,1), round(r[3]/1e3,2), round(r[4],2), r[0][:100])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print(cols)
rows = list(con.execute("select name, start, end from kernels order by start"))
plan = [k for k, r in enumerate(rows) if "plan" in r[0]]
# one epoch in the middle: from one plan-kernel group to the next
starts = [k for n, k in enumerate(plan) if n == 0 or plan[n] - plan[n - 1] > 3]
a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = rows[a][1]
for name, s, e in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {name[:90]}")
try:
    mc = list(con.execute("select * from memory_copies limit 3")); print(mc)
except Exception as ex: print("no memcpy view", ex)
PY
