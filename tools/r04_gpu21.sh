#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/probe_config4_rank_share.py 2> $O/r04_c4rank.err | tee $O/r04_config4_rank_share.jsonl; tail -3 $O/r04_c4rank.err
cat > /tmp/plan21.json <<'PLAN'
[{"dataset":"xl6m-clustered","lr0":0.05,"seed":7,"mode":"item-deferred","epochs":12,"eval_every":3,"world":2,"layout":"replicated"},
 {"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item-deferred","epochs":40,"eval_every":5,"world":4,"layout":"replicated"},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred","epochs":20,"eval_every":5,"world":4,"layout":"replicated"}]
PLAN
timeout 900 python tools/paired_recall.py $O/r04_paired_plan21.json /tmp/plan21.json > $O/r04_paired_plan21.log 2>&1; echo "plan21 exit $?"; grep -v "^{" $O/r04_paired_plan21.log | tail -3; grep "^{" $O/r04_paired_plan21.log | cut -c1-300
