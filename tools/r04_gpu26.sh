#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_tf_golden.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -k "simgcl or info_nce or SimGCL" > $O/r04_pytest_i.log 2>&1; echo "pytest exit $?"; tail -6 $O/r04_pytest_i.log | cut -c1-250
for k in 1 2 3; do python tools/bench_eval_simgcl.py --skip-eval 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('simgcl ms/step', round(d['simgcl']['ms_per_step'],4))"; done
