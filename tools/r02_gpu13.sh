#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QREC_EVAL_EXPERIMENT_INF_TAU=1 REPS=2 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_eval_inf -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval_inf.log 2>&1
python - <<PY
import sqlite3
con=sqlite3.connect("$O/prof_eval_inf/r02_results.db")
for r in con.execute("select name,total_calls,average from top_kernels where name like '%score_filter%'"):
    print("inf tau: %3d calls %8.3f ms avg  %s"%(r[1],r[2]/1e3,r[0][:70]))
PY
