#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_bpr.py -q -m gpu -x > $O/t_bpr.log 2>&1; echo "bpr tests exit $?"; tail -6 $O/t_bpr.log
timeout 600 python tools/probe_exact.py > $O/probe_exact.json 2> $O/probe_exact.err; echo "probe exit $?"; grep -E "^f(64|32) (1|4|6|8|12) " $O/probe_exact.json | cut -c1-330
