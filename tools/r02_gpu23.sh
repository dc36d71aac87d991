#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dist.py -q -x > $O/t_graph.log 2>&1; echo "graph+dist tests exit $?"; tail -3 $O/t_graph.log | head -2
python tools/bench_lightgcn.py --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lightgcn', round(d['spmm_ms'],4), round(d['ms_per_step'],4))"
python tools/bench_eval_simgcl.py --skip-eval 2>/dev/null | cut -c1-60
python tools/prof_ngcf.py 2>/dev/null | tail -1
