#!/bin/bash
# SpMM locality matrix: LightGCN step / SpMM time by graph (structureless vs planted communities) and QREC_SPMM_CHUNKS
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/spmm_matrix.jsonl
for shape in yelp2018 yelp2018-clustered; do
  for c in 1 2 4; do
    QREC_SPMM_CHUNKS=$c timeout 300 python tools/bench_lightgcn.py --steps 60 --shape $shape >> $O/spmm_matrix.jsonl 2>$O/spmm_err.log || tail -3 $O/spmm_err.log
  done
done
python - <<'P'
import json
for l in open('/root/repo/gpurun_out/spmm_matrix.jsonl'):
    d=json.loads(l); print(d['workload'][30:62], 'chunks',d['spmm_chunks'],'spmm_ms',round(d['spmm_ms'],4),'step',round(d['ms_per_step'],4))
P
