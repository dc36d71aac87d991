#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_bpr.py -q -x > $O/t_bpr.log 2>&1; echo "bpr tests exit $?"; tail -4 $O/t_bpr.log
python bench.py --no-cpu-baseline > $O/bench_exact.json 2>$O/bench_exact.err; python -c "
import json; d=json.load(open('$O/bench_exact.json')); print(json.dumps(d['exact_mode']))"
