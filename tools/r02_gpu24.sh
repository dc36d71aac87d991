#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_sgl
rocprofv3 --kernel-trace --stats -d $O/prof_sgl -o sgl -- python $R/tools/bench_sgl.py > $O/prof_sgl.log 2>&1; echo "exit $?"; grep -o '"ms_per_step": [0-9.]*' $O/prof_sgl.log
python - <<'P'
import sqlite3
con=sqlite3.connect('/root/repo/gpurun_out/prof_sgl/sgl_results.db')
for name,calls,t,avg,pct in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:18]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:80]}")
P
