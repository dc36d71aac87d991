#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_graph.py -q -k "unique_per_batch or simgcl or throughput" > $O/t_sg.log 2>&1; echo "tests exit $?"; tail -4 $O/t_sg.log
QREC_MODE=throughput QREC_BENCH_EPOCHS=2 python tools/bench_class_epoch.py SimGCL 2>/dev/null | cut -c1-300
QREC_MODE=throughput QREC_BENCH_EPOCHS=6 python tools/bench_class_epoch.py SimGCL 2>/dev/null | cut -c1-300
