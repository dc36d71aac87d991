#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -q -x -k "ngcf or row_subset" > $O/t_ngcf.log 2>&1; echo "ngcf tests exit $?"; tail -5 $O/t_ngcf.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_ngcf
rocprofv3 --kernel-trace --stats -d $O/prof_ngcf -o ngcf -- python $R/tools/prof_ngcf.py > $O/prof_ngcf.log 2>&1; echo "exit $?"; grep ms_per_step $O/prof_ngcf.log
python - <<'P'
import sqlite3
con=sqlite3.connect('/root/repo/gpurun_out/prof_ngcf/ngcf_results.db')
for name,calls,t,avg,pct in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:18]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:70]}")
P
