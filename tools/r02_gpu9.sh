#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "throughput_mode_of_the_pairwise" > $O/t_a.log 2>&1; echo "pairwise throughput exit $?"; tail -12 $O/t_a.log
timeout 900 python -m pytest tests/test_gpu_bpr.py -q -m gpu -k "cross_validation" > $O/t_b.log 2>&1; echo "cv exit $?"; tail -8 $O/t_b.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "ALL gpu tests exit $?"; tail -8 $O/t_all.log
