#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1 2; do
python tools/bench_lightgcn.py --steps 60 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['spmm_ms'],4), round(d['ms_per_step'],4))"
QREC_SPMM_NT=1 python tools/bench_lightgcn.py --steps 60 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt   ', round(d['spmm_ms'],4), round(d['ms_per_step'],4))"
done
