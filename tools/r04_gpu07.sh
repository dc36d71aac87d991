#!/bin/bash
# round 4, call 7: targeted tests after the K = G default and the row-partitioned batch rows; then the full paired-Recall ledger
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider -x -k "row_partitioned or bench or logical or throughput_mode_of" > $O/r04_pytest_c.log 2>&1; echo "pytest dist/graph exit $?"; tail -5 $O/r04_pytest_c.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_bpr.py -m gpu -q -p no:cacheprovider -k "structured or reconciliation or auto_schedule or whole_item_runs or two_ranks" > $O/r04_pytest_d.log 2>&1; echo "pytest bpr exit $?"; tail -8 $O/r04_pytest_d.log | cut -c1-250
timeout 1500 python tools/paired_recall.py $O/r04_paired_recall.json full > $O/r04_paired_full.log 2>&1; echo "ledger exit $?"; grep -v "^{" $O/r04_paired_full.log | tail -3; grep -c "^{" $O/r04_paired_full.log
