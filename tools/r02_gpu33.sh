#!/bin/bash
# eval: tests in the three routes (bf16 default, fp32 filter, bf16 with a sparse sample), fused timing, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_eval.py -q -x > $O/t_eval.log 2>&1; echo "eval tests exit $?"; tail -1 $O/t_eval.log
QREC_EVAL_F32_FILTER=1 timeout 1500 python -m pytest tests/test_gpu_eval.py -q -x > $O/t_eval_f32.log 2>&1; echo "eval tests (fp32 filter) exit $?"; tail -1 $O/t_eval_f32.log
QREC_EVAL_BF16_STRIDE=8 timeout 1500 python -m pytest tests/test_gpu_eval.py -q -x > $O/t_eval_s8.log 2>&1; echo "eval tests (bf16, stride 8) exit $?"; tail -1 $O/t_eval_s8.log
python tools/bench_eval.py > $O/eval.json 2>$O/eval.err || tail -5 $O/eval.err; cut -c1-230 $O/eval.json
QREC_EVAL_F32_FILTER=1 python tools/bench_eval.py > $O/eval_f32_filter.json 2>$O/eval.err || tail -5 $O/eval.err; cut -c1-60 $O/eval_f32_filter.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_eval
rocprofv3 --kernel-trace --stats -d $O/prof_eval -o eval -- python $R/tools/bench_eval.py child > $O/prof_eval.log 2>&1; echo "exit $?"
python - <<'P' | tee /root/repo/gpurun_out/eval_kernel_stats.txt
import sqlite3
con=sqlite3.connect('/root/repo/gpurun_out/prof_eval/eval_results.db')
print(f"{'calls':>6} {'total us':>10} {'avg us':>9} {'%':>6}  kernel")
for name,calls,t,avg,pct in list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:16]:
    print(f"{calls:6d} {t/1e3:10.1f} {avg/1e3:9.2f} {pct:6.2f}  {name[:110]}")
P
