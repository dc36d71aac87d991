#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_graph.py -q -x -k "buir or graph_models_data_parallel" > $O/t_buir.log 2>&1; echo "buir tests exit $?"; tail -3 $O/t_buir.log | head -2
python tools/prof_buir.py 2>/dev/null | tail -1
