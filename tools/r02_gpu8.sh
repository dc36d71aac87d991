#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 2 0; do
  QREC_EVAL_VARIANT=$v REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_eval_v$v -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval_v$v.log 2>&1
  python - <<PY
import sqlite3
con=sqlite3.connect("$O/prof_eval_v$v/r02_results.db")
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels where name like '%score_filter%' or name like '%select_topk%'"):
    print("variant $v: %6d %10.3f ms avg  %s"%(r[1],r[3]/1e6,r[0][:70]))
PY
  tail -1 $O/prof_eval_v$v.log | cut -c1-200
done
