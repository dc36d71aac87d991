#!/bin/bash
# kernel timeline of one epoch of the sharded layout at world 1 (default flags) under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_shard
QREC_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_shard -o shard -- python $R/bench.py --dist-mode sharded --steps 3 --warmup 1 --epochs-per-step 20 --no-cpu-baseline --no-extras > $O/prof_shard.log 2>&1; echo "exit $?"
cd $R
python - <<'PY' > gpurun_out/r03_sharded_epoch_timeline.txt
import sqlite3, glob
db = glob.glob("gpurun_out/prof_shard/*results.db")[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, stream_id from kernels order by start"))
plan = [k for k, r in enumerate(rows) if "plan_mark" in r[0]]
a, b = plan[len(plan) // 2], plan[len(plan) // 2 + 1]
t0 = rows[a][1]
print("# one epoch of `QREC_FORCE_DIST=1 python bench.py --dist-mode sharded` (world 1, default flags: 2 batches, plan begun in front of the last batch) under rocprofv3 --kernel-trace")
print("# from one plan_mark launch to the next;  columns: start_us  duration_us  stream  kernel")
for name, s, e, st in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} s{st} {name[:100]}")
PY
cat gpurun_out/r03_sharded_epoch_timeline.txt | cut -c1-150
