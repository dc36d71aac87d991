#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bpr.py -m gpu -q -x -p no:cacheprovider -k "ordered or scheduled or exact or bpr_model_end_to_end or pipelined" > $O/r03_exact_tests.log 2>&1
echo "exact tests exit $?"; tail -8 $O/r03_exact_tests.log | cut -c1-220
PROBE_KERNELS=g16 timeout 300 python tools/probe_exact.py > $O/r03_exact_probe2.log 2>&1; echo "probe exit $?"; grep -v "^{" $O/r03_exact_probe2.log | cut -c1-200 | tail -12
timeout 300 python tools/probe_exact_dbg.py > $O/r03_exact_dbg2.log 2>&1; echo "dbg probe exit $?"; grep -v "^{" $O/r03_exact_dbg2.log | tail -24
timeout 120 tools/ubench/atomics4 > $O/r03_ubench_atomics4.txt 2>&1; echo "ubench exit $?"; cat $O/r03_ubench_atomics4.txt
