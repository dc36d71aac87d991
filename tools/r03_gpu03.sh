#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python tools/probe_exact_dbg.py > $O/r03_exact_dbg.log 2>&1; echo "dbg probe exit $?"; grep -v "^{" $O/r03_exact_dbg.log | tail -30
timeout 600 python -m pytest tests/test_gpu_bpr.py -m gpu -q -p no:cacheprovider -k "ordered or scheduled or exact or bpr_model_end_to_end or pipelined or basicmf or pmf or svdpp or tbpr or mf_family or cross_validation or main_flow" > $O/r03_exact_tests.log 2>&1
echo "exact tests exit $?"; tail -8 $O/r03_exact_tests.log | cut -c1-220
