#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bpr.py -m gpu -q -p no:cacheprovider -k "reconciliation_batches or two_ranks" > $O/r04_pytest_g.log 2>&1; echo "pytest exit $?"; tail -15 $O/r04_pytest_g.log | cut -c1-250
