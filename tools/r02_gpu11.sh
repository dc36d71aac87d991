#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_graph.py -q -x -k "spmm or lightgcn" > $O/t_spmm.log 2>&1; echo "spmm tests exit $?"; tail -3 $O/t_spmm.log
: > $O/spmm_matrix.jsonl
for shape in yelp2018 yelp2018-clustered; do
  for pc in "4 1" "8 1" "4 4" "8 4"; do
    set -- $pc
    QREC_SPMM_DEEP=$1 QREC_SPMM_CHUNKS=$2 timeout 300 python tools/bench_lightgcn.py --steps 60 --shape $shape | sed "s/^{/{\"deep\": $1, /" >> $O/spmm_matrix.jsonl 2>$O/spmm_err.log || tail -3 $O/spmm_err.log
  done
done
python - <<'P'
import json
for l in open('/root/repo/gpurun_out/spmm_matrix.jsonl'):
    d=json.loads(l); print(d['workload'][30:60], 'deep',d['deep'],'chunks',d['spmm_chunks'],'spmm_ms',round(d['spmm_ms'],4),'step',round(d['ms_per_step'],4))
P
