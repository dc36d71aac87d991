#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/probe_hbm_resident_deferred.py '[{"S":4},{"S":4,"groups":16384},{"S":4,"groups":4096},{"S":4,"flush":32},{"S":4,"item_run":32},{"S":4,"item_run":0},{"S":2},{"S":8},{"S":4,"sub_chunk":48},{"S":4,"sub_chunk":64,"flush":32},{"S":1,"chunk":32},{"schedule":"item","S":1},{"schedule":"user","S":1}]' 2>&1 | tee $O/r04_hbm_resident_deferred.jsonl | cut -c1-200
python - > /tmp/plan12.json <<'PY'
import json
c=[]
for lr0,ep in ((0.01,25),(0.05,15)):
    for mode in ("item","user","item-deferred"):
        c.append(dict(dataset="yelp2018",lr0=lr0,seed=7,mode=mode,epochs=ep,eval_every=5 if lr0==0.01 else 3))
Y="yelp2018-clustered"
for world,layout in ((2,"replicated"),(4,"replicated"),(4,"sharded")):
    c.append(dict(dataset=Y,lr0=0.01,seed=11,mode="item",epochs=40,eval_every=5,world=world,layout=layout))
c.append(dict(dataset=Y,lr0=0.01,seed=7,mode="user",epochs=40,eval_every=5,world=4,layout="replicated"))
print(json.dumps(c))
PY
timeout 900 python tools/paired_recall.py $O/r04_paired_plan12.json /tmp/plan12.json > $O/r04_paired_plan12.log 2>&1; echo "plan12 exit $?"; grep "^{" $O/r04_paired_plan12.log | cut -c1-330
