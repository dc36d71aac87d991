#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider -k "row_partitioned or row_subset" > $O/r04_pytest_f.log 2>&1; echo "pytest exit $?"; tail -12 $O/r04_pytest_f.log | cut -c1-250
