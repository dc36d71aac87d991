#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
run() { # name, flags
  QREC_FORCE_DIST=1 MASTER_PORT=29611 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras $2 > $O/r03_shard_$1.json 2> $O/r03_shard_$1.err
  python - <<PY
import json
try:
    d = json.load(open("$O/r03_shard_$1.json"))
    print("$1", "ms/epoch", round(d["config"]["ms_per_epoch"], 4), "batches", d["config"]["batches_per_epoch"], "piped", d["config"]["fetch_pipelined"], "ahead", d["config"]["plan_ahead"], "loss", round(d["config"]["final_loss"]))
except Exception as e:
    print("$1 failed", e); print(open("$O/r03_shard_$1.err").read()[-1500:])
PY
}
export GPU_MAX_HW_QUEUES=8
run q8_base "--no-shard-pipeline --no-plan-ahead"
run q8_ahead "--no-shard-pipeline"
run q8_both ""
run q8_ahead_1batch "--no-shard-pipeline --shard-batch 2097152"
run q8_both_2batch "--shard-batch 700000"
unset GPU_MAX_HW_QUEUES
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_shard
QREC_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_shard -o shard -- python $R/bench.py --dist-mode sharded --no-shard-pipeline --shard-batch 2097152 --steps 3 --warmup 1 --epochs-per-step 20 --no-cpu-baseline --no-extras > $O/prof_shard.log 2>&1; echo "exit $?"
cd $R
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/prof_shard/*results.db")[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
plan = [k for k, r in enumerate(rows) if "plan_mark" in r[0]]
a, b = plan[len(plan) // 2], plan[len(plan) // 2 + 1]
t0 = rows[a][1]
for name, s, e, st, q in rows[a:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} s{st} q{q} {name[:70]}")
PY
