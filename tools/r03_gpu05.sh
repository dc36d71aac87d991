#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bpr.py -m gpu -q -x -p no:cacheprovider -k "ordered or scheduled or exact or bpr_model_end_to_end or pipelined" > $O/r03_exact_tests.log 2>&1
echo "exact tests exit $?"; tail -12 $O/r03_exact_tests.log | cut -c1-220
timeout 300 python tools/probe_exact.py > $O/r03_exact_probe3.log 2>&1; echo "probe exit $?"; grep -v "^{" $O/r03_exact_probe3.log | cut -c1-200 | tail -14
timeout 300 python tools/probe_exact_dbg.py > $O/r03_exact_dbg3.log 2>&1; echo "dbg probe exit $?"; grep -v "^{" $O/r03_exact_dbg3.log | tail -16
