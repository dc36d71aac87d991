#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_graph.py -q -x > $O/t_graph.log 2>&1; echo "graph tests exit $?"; tail -5 $O/t_graph.log
python tools/bench_eval_simgcl.py --skip-eval > $O/simgcl.json 2> $O/simgcl.err || tail -5 $O/simgcl.err; cat $O/simgcl.json | cut -c1-600
python tools/bench_lightgcn.py --steps 60 | cut -c1-400
