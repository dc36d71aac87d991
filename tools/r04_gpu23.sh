#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for c in 0 32 48 64; do
  timeout 200 python bench.py --chunk $c --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk=$c ->', d['config']['chunk'], 'ms/epoch', round(d['config']['ms_per_epoch'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))"
done 2>&1 | tee $O/r04_chunk_vs_runs.txt
