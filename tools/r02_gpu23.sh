#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_graph.py -q -x -k "sgl or graph_models_data_parallel" > $O/t_sgl.log 2>&1; echo "sgl tests exit $?"; tail -3 $O/t_sgl.log | head -2
python tools/bench_sgl.py 2>/dev/null | cut -c1-140
