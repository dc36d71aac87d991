#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/prof_buir.py
QREC_SPMM_CHUNKS=1 python $R/tools/prof_buir.py
rm -rf $O/prof_buir
rocprofv3 --kernel-trace --stats -d $O/prof_buir -o buir -- python $R/tools/prof_buir.py > $O/prof_buir.log 2>&1; echo "exit $?"
python - <<'P'
import sqlite3
con=sqlite3.connect('/root/repo/gpurun_out/prof_buir/buir_results.db')
for name,calls,t,avg,pct in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:14]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:80]}")
P
