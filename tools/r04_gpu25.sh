#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -k "hot_row" > $O/r04_pytest_h.log 2>&1; echo "pytest exit $?"; tail -12 $O/r04_pytest_h.log | cut -c1-250
