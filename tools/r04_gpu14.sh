#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/probe_strong_scaling_bound.py yelp2018 > $O/r04_strong_scaling_bound.json 2> $O/r04_ssb.err; echo "exit $?"; tail -3 $O/r04_ssb.err; python -c "
import json; d=json.load(open('$O/r04_strong_scaling_bound.json'))
for k,v in d['ranks'].items(): print(k, round(v['ms_per_epoch_no_links'],4), 'ceil', round(v['speedup_ceiling_no_links'],2), {a:round(b,2) for a,b in v['speedup_with_link_arithmetic'].items()}, 'wire MB', round(v['ring_wire_MB_per_rank_per_epoch'],1))"
timeout 600 python -m pytest tests/test_gpu_bpr.py -m gpu -q -p no:cacheprovider -k "two_ranks" 2>&1 | tail -2
