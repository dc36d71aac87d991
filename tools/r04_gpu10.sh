#!/bin/bash
# round 4, call 10: reconcile only the HOT item rows inside the epoch (G - 1 times) + the whole table at its close: how few rows keep the Recall bar?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
python - > /tmp/plan10.json <<'PY'
import json
c=[]
Y="yelp2018-clustered"
for lr0,ep in ((0.01,40),(0.05,20)):
    for world in (4,8):
        for H in (1024,4096,12288):
            c.append(dict(dataset=Y,lr0=lr0,seed=7,mode="item",epochs=ep,eval_every=5,world=world,layout="replicated",hot_rows=H))
print(json.dumps(c))
PY
timeout 1500 python tools/paired_recall.py $O/r04_paired_plan10.json /tmp/plan10.json > $O/r04_paired_plan10.log 2>&1; echo "plan10 exit $?"; grep -v "^{" $O/r04_paired_plan10.log | tail -5; grep "^{" $O/r04_paired_plan10.log | cut -c1-330
