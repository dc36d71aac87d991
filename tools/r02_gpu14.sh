#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_graph.py -q -x -k "simgcl or nce or sgl or sept or contrast" > $O/t_graph.log 2>&1; echo "graph tests exit $?"; tail -3 $O/t_graph.log
for k in 1 2 3; do python tools/bench_eval_simgcl.py --skip-eval 2>/dev/null | cut -c1-120; done
