#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_eval.py -q -m gpu -x > $O/t_eval.log 2>&1; echo "eval tests exit $?"; tail -4 $O/t_eval.log
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_eval_v3 -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval_v3.log 2>&1
python - <<PY
import sqlite3
con=sqlite3.connect("$O/prof_eval_v3/r02_results.db")
tot=0
for r in con.execute("select name,total_calls,average from top_kernels order by total_duration desc"):
    if r[1] in (5,10): tot += r[2]*r[1]/5
    print("%3d calls %8.3f ms avg  %s"%(r[1],r[2]/1e3,r[0][:70]))
print("kernel time per evaluation %.3f ms"%(tot/1e3))
PY
grep gpu_ms $O/prof_eval_v3.log | cut -c1-140
REPS=1 timeout 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/prof_eval_mfma -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval_mfma.log 2>&1
python - <<PY
import sqlite3
con=sqlite3.connect("$O/prof_eval_mfma/r02_results.db")
for r in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='MfmaUtil' and kernel_name like '%score%' group by kernel_name"):
    print("MfmaUtil %3d %8.2f  %s"%(r[1],r[2],r[0][:80]))
PY
