#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_eval.py -q -m gpu > $O/t_eval.log 2>&1; echo "eval tests exit $?"; tail -5 $O/t_eval.log
timeout 600 python tools/bench_eval.py > $O/bench_eval_r02.json 2> $O/bench_eval_r02.err; echo "bench eval exit $?"; cat $O/bench_eval_r02.json; tail -3 $O/bench_eval_r02.err
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_eval -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval.log 2>&1; echo "prof exit $?"
REPS=2 timeout 600 rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/prof_eval_mfma -o r02 -- python $R/tools/bench_eval.py child > $O/prof_eval_mfma.log 2>&1; echo "pmc exit $?"
python - <<PY
import sqlite3
con=sqlite3.connect("$O/prof_eval/r02_results.db")
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%6d %12.1f %10.3f %6.2f  %s"%(r[1],r[2]/1e3,r[3]/1e3,r[4],r[0][:100]))
try:
    con=sqlite3.connect("$O/prof_eval_mfma/r02_results.db")
    for r in con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='MfmaUtil' group by kernel_name"):
        print("MfmaUtil %6d %8.2f  %s"%(r[1],r[2],r[0][:100]))
except Exception as e: print("pmc read failed", e)
PY
