#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for H in 0 4096; do
PROBE_HOT_ROWS=$H timeout 600 python tools/probe_strong_scaling_bound.py yelp2018 > $O/r04_strong_scaling_bound_h$H.json 2> $O/r04_ssb.err; echo "H=$H exit $?"; tail -3 $O/r04_ssb.err; python -c "
import json; d=json.load(open('$O/r04_strong_scaling_bound_h$H.json'))
for k,v in d['ranks'].items(): print(k, round(v['ms_per_epoch_no_links'],4), 'ceil', round(v['speedup_ceiling_no_links'],2), {a:round(b,2) for a,b in v['speedup_with_link_arithmetic'].items()}, 'wire MB', round(v['ring_wire_MB_per_rank_per_epoch'],1))"
done
cat > /tmp/plan15.json <<'PLAN'
[{"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item","epochs":40,"eval_every":5,"world":8,"layout":"replicated"},
 {"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item","epochs":40,"eval_every":5,"world":8,"layout":"sharded"},
 {"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item","epochs":40,"eval_every":5,"world":4,"layout":"replicated"},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":8,"layout":"replicated"}]
PLAN
timeout 600 python tools/paired_recall.py $O/r04_paired_plan15.json /tmp/plan15.json > $O/r04_paired_plan15.log 2>&1; echo "plan15 exit $?"; grep "^{" $O/r04_paired_plan15.log | cut -c1-300
