#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_dist.py -q -x -k "row_partitioned or simgcl" > $O/t_rp.log 2>&1; echo "dist row-partition tests exit $?"; tail -15 $O/t_rp.log
timeout 1500 python -m pytest tests/test_gpu_graph.py -q -k "row_partitioned or simgcl" > $O/t_rp2.log 2>&1; echo "graph row-partition tests exit $?"; tail -15 $O/t_rp2.log
