#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_graph.py -m gpu -q -x -p no:cacheprovider -k "spmm" > $O/r03_spmm_tests.log 2>&1; echo "spmm tests exit $?"; tail -4 $O/r03_spmm_tests.log | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_bpr.py -m gpu -q -x -p no:cacheprovider -k "scheduled or pipelined" > $O/r03_exact_tests2.log 2>&1; echo "exact tests exit $?"; tail -4 $O/r03_exact_tests2.log | cut -c1-200
for shape in yelp2018 yelp2018-clustered; do for cs in 1 2; do
  QREC_SPMM_COLSPLIT=$cs timeout 200 python tools/bench_lightgcn.py --shape $shape --steps 60 > $O/r03_lightgcn_${shape}_cs$cs.json 2> $O/r03_lightgcn_${shape}_cs$cs.err
  echo "$shape colsplit $cs: $(python -c "import json,sys; d=json.load(open('$O/r03_lightgcn_${shape}_cs$cs.json')); print('spmm_ms', round(d['spmm_ms'],4), 'step_ms', round(d['ms_per_step'],4), 'chunks', d['spmm_chunks'], 'segs', d['segments'], 'long', d['long_rows'])")"
done; done
