#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
grep -n -E "GRBM_GUI_ACTIVE|MfmaUtil" -A3 $O/counters_avail.txt | head -30
REPS=1 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES --kernel-trace -d $O/pmc_gui -o r02 -- python $R/tools/bench_eval.py child > $O/pmc_gui.log 2>&1; echo "pmc exit $?"
python - <<PY
import sqlite3
con=sqlite3.connect("$O/pmc_gui/r02_results.db")
for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%score_filter%' group by kernel_name, counter_name"):
    print("   %-24s n=%d avg=%.5g"%(r[1],r[2],r[3]))
try:
    for r in con.execute("select name, average from top_kernels where name like '%score_filter%'"): print("   avg duration us", r[1])
except Exception as e:
    cols=[r for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    print(cols[:40])
PY
