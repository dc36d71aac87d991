#!/bin/bash
# round 4: BASELINE config #4 at FULL size on one GPU (10 M x 1 M, d = 128, 200 M triplets per epoch) under the schedule `auto` picks there
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1100 python tools/bench_config4_full.py auto > $O/r04_config4_full_single_gpu.json 2> $O/r04_c4.err; echo "exit $?"; tail -3 $O/r04_c4.err; cut -c1-900 $O/r04_config4_full_single_gpu.json
