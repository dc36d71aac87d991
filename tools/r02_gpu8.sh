#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_eval.py -q -m gpu -x > $O/t_eval.log 2>&1; echo "eval tests exit $?"; tail -4 $O/t_eval.log
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
  QREC_EVAL_VARIANT=$v REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_eval_v$v -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval_v$v.log 2>&1
  python - <<PY
import sqlite3
con=sqlite3.connect("$O/prof_eval_v$v/r02_results.db")
tot=0
for r in con.execute("select name,total_calls,average from top_kernels order by total_duration desc"):
    if r[1] in (5,10): tot += r[2]*r[1]/5
    if 'score_filter' in r[0] or 'select_topk' in r[0] or 'exact_wave' in r[0]: print("variant $v: %3d calls %8.3f ms avg  %s"%(r[1],r[2]/1e3,r[0][:70]))
print("variant $v: kernel time per evaluation %.3f ms"%(tot/1e3))
PY
  grep gpu_ms $O/prof_eval_v$v.log | cut -c1-120
done
