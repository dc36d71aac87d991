#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bpr.py -m gpu -q -s -p no:cacheprovider -k "bpr_class_on_two_ranks" > $O/r03_t30.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|final loss|Error|error" $O/r03_t30.log | cut -c1-300 | tail -12
