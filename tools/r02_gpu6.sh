#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_eval.py -q -m gpu -x > $O/t_eval.log 2>&1; echo "eval tests exit $?"; tail -15 $O/t_eval.log
timeout 600 python tools/bench_eval_simgcl.py > $O/eval_simgcl_r02.json 2> $O/eval_simgcl_r02.err; echo "bench eval exit $?"; tail -c 1500 $O/eval_simgcl_r02.json; tail -3 $O/eval_simgcl_r02.err
timeout 900 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "row_partitioned" > $O/t_graph_dp.log 2>&1; echo "graph rowpart exit $?"; tail -5 $O/t_graph_dp.log
