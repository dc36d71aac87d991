#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -q -x -k "spmm or lightgcn" > $O/t_spmm.log 2>&1; echo "spmm tests exit $?"; tail -3 $O/t_spmm.log
for k in 1 2; do
python tools/bench_lightgcn.py --steps 60 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('yelp      ', round(d['spmm_ms'],4), round(d['ms_per_step'],4))"
python tools/bench_lightgcn.py --steps 60 --shape yelp2018-clustered | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clustered ', round(d['spmm_ms'],4), round(d['ms_per_step'],4))"
done
