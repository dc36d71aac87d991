#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_simgcl
rocprofv3 --kernel-trace --stats -d $O/prof_simgcl -o simgcl -- python $R/tools/bench_eval_simgcl.py --skip-eval > $O/prof_simgcl.log 2>&1; echo "exit $?"
python - <<'P'
import sqlite3
con=sqlite3.connect('/root/repo/gpurun_out/prof_simgcl/simgcl_results.db')
for name,calls,t,avg,pct in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:24]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:70]}")
P
