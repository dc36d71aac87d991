#!/bin/bash
# round 4: refreshed kernel stats of the other configs' steps (LightGCN config #3, NGCF / SimGCL config #5, evaluation) and one more default bench run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in lightgcn simgcl ngcf eval; do
  rm -rf $O/prof_$m
  case $m in
    lightgcn) cmd="python $R/tools/bench_lightgcn.py --steps 60";;
    simgcl) cmd="python $R/tools/bench_eval_simgcl.py --skip-eval";;
    ngcf) cmd="python $R/tools/prof_ngcf.py";;
    eval) cmd="python $R/tools/bench_eval.py child";;
  esac
  rocprofv3 --kernel-trace --stats -d $O/prof_$m -o $m -- $cmd > $O/prof_$m.log 2>&1; echo "$m exit $?"
  db=$(ls $O/prof_$m/*_results.db $O/prof_$m/*/*_results.db 2>/dev/null | head -1)
  python $R/tools/summarize_stats.py $db $O/r04_${m}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $cmd   [r04, Yelp2018 shape d=64; profiled run]" | head -8
done
cd $R
timeout 600 python bench.py > $O/r04_bench_again.json 2> $O/r04_bench_again.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$O/r04_bench_again.json')); print(d['value'], d['roofline']['frac'], [round(x['abs_diff'],5) for x in d['recall_at_20']['datasets']])"
