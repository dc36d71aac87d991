#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider > $O/r03_dist_tests2.log 2>&1; echo "dist tests exit $?"; tail -5 $O/r03_dist_tests2.log | cut -c1-250
QREC_GRAPH_EXCHANGE=referenced timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -q -x -p no:cacheprovider -k "row_partitioned" > $O/r03_graph_rows_ref.log 2>&1; echo "class rows referenced exit $?"; tail -4 $O/r03_graph_rows_ref.log | cut -c1-250
