#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu > $O/t_dist.log 2>&1; echo "dist tests exit $?"; tail -12 $O/t_dist.log
timeout 900 python -m pytest tests/test_gpu_graph.py -q -m gpu -k "row_partitioned or data_parallel_two_ranks" > $O/t_graph_dp.log 2>&1; echo "graph dp tests exit $?"; tail -12 $O/t_graph_dp.log
timeout 900 python -m pytest tests/test_gpu_bpr.py -q -m gpu -k "two_ranks or cross_validation" > $O/t_bpr_dp.log 2>&1; echo "bpr dp tests exit $?"; tail -8 $O/t_bpr_dp.log
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', round(d['value']/1e9,3), 'G/s', round(d['config']['ms_per_epoch'],4), 'ms/epoch', round(d['roofline']['avg_launch_ms'],4), 'kernel ms', d['config']['epochs_per_step'], d['config'].get('final_loss'))"; }
A="--no-cpu-baseline --no-extras"
QREC_FORCE_DIST=1 timeout 300 python bench.py $A > $O/f1.json 2>/dev/null; show $O/f1.json "force replicated"
QREC_FORCE_DIST=1 timeout 300 python bench.py $A --dist-mode sharded > $O/f2.json 2>/dev/null; show $O/f2.json "force sharded"
timeout 300 python bench.py $A > $O/f5.json 2>/dev/null; show $O/f5.json "plain"
