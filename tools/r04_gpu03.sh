#!/bin/bash
# round 4, call 3: N = 2 / 4 logical ranks again (ThreadComm.allreduce now waits for the rank's stream), both layouts, both datasets
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
python - > /tmp/plan3.json <<'PY'
import json
c=[]
Y="yelp2018-clustered"
for mode in ("item","user"):
    for world in (2,4):
        for K in (1,4):
            c.append(dict(dataset=Y,lr0=0.05,seed=7,mode=mode,epochs=20,eval_every=5,world=world,layout="replicated",syncs=K))
for world in (2,4):
    c.append(dict(dataset=Y,lr0=0.01,seed=7,mode="item",epochs=40,eval_every=5,world=world,layout="replicated"))
    c.append(dict(dataset=Y,lr0=0.01,seed=7,mode="item",epochs=40,eval_every=5,world=world,layout="sharded"))
    for lr0,ep in ((0.05,20),(0.01,40)):
        c.append(dict(dataset="lastfm",lr0=lr0,seed=7,mode="item",epochs=ep,eval_every=4,world=world,layout="replicated"))
        c.append(dict(dataset="lastfm",lr0=lr0,seed=7,mode="item",epochs=ep,eval_every=4,world=world,layout="sharded",shard_batch=16384))
c.append(dict(dataset=Y,lr0=0.05,seed=7,mode="item",epochs=20,eval_every=5,world=8,layout="replicated"))
c.append(dict(dataset=Y,lr0=0.05,seed=7,mode="item",epochs=20,eval_every=5,world=8,layout="sharded"))
print(json.dumps(c))
PY
timeout 1200 python tools/paired_recall.py $O/r04_paired_plan3.json /tmp/plan3.json > $O/r04_paired_plan3.log 2>&1; echo "plan3 exit $?"; grep -v "^{" $O/r04_paired_plan3.log | tail -5; grep "^{" $O/r04_paired_plan3.log | cut -c1-330
