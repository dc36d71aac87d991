#!/bin/bash
# round 4, call 2: user-major vs item-major on structured data; own-order sequential statement; replicated layout with K syncs per epoch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
python - > /tmp/plan2.json <<'PY'
import json
c=[]
Y="yelp2018-clustered"
for ds,runs,ev in ((Y,((0.05,20),(0.01,40)),5),("lastfm",((0.05,20),(0.01,40)),2)):
    for lr0,ep in runs:
        c.append(dict(dataset=ds,lr0=lr0,seed=7,mode="user",epochs=ep,eval_every=ev))
        c.append(dict(dataset=ds,lr0=lr0,seed=7,mode="item",epochs=ep,eval_every=ev,own_order=True))
for mode in ("user","item"):
    for world in (2,4):
        for K in (1,2,4,8):
            c.append(dict(dataset=Y,lr0=0.05,seed=7,mode=mode,epochs=20,eval_every=5,world=world,layout="replicated",syncs=K))
        c.append(dict(dataset=Y,lr0=0.05,seed=7,mode=mode,epochs=20,eval_every=5,world=world,layout="sharded"))
print(json.dumps(c))
PY
timeout 1200 python tools/paired_recall.py $O/r04_paired_plan2.json /tmp/plan2.json > $O/r04_paired_plan2.log 2>&1; echo "plan2 exit $?"; grep -v "^{" $O/r04_paired_plan2.log | tail -5; grep "^{" $O/r04_paired_plan2.log | cut -c1-330
