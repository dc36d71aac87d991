#!/bin/bash
# kernel stats of the three graph-model steps (LightGCN config #3, SimGCL / NGCF config #5), Yelp2018 shape
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in lightgcn simgcl ngcf; do
  rm -rf $O/prof_$m
  case $m in
    lightgcn) cmd="python $R/tools/bench_lightgcn.py --steps 60";;
    simgcl) cmd="python $R/tools/bench_eval_simgcl.py --skip-eval";;
    ngcf) cmd="python $R/tools/prof_ngcf.py";;
  esac
  rocprofv3 --kernel-trace --stats -d $O/prof_$m -o $m -- $cmd > $O/prof_$m.log 2>&1; echo "$m exit $?"; grep -o '"\?ms_per_step"\?:\? [0-9.]*' $O/prof_$m.log | head -1
done
