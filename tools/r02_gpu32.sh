#!/bin/bash
# eval: tests (default route), fused timing, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_eval.py -q -x > $O/t_eval.log 2>&1; echo "eval tests exit $?"; tail -3 $O/t_eval.log
python tools/bench_eval.py > $O/eval.json 2>$O/eval.err || tail -5 $O/eval.err; cut -c1-230 $O/eval.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_eval
rocprofv3 --kernel-trace --stats -d $O/prof_eval -o eval -- python $R/tools/bench_eval.py child > $O/prof_eval.log 2>&1; echo "exit $?"
python - <<'P'
import sqlite3
con=sqlite3.connect('/root/repo/gpurun_out/prof_eval/eval_results.db')
for name,calls,t,avg,pct in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:14]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:80]}")
P
