#!/bin/bash
# round 4, call 6: how often must the ranks reconcile?  N = 2 / 4 / 8 logical ranks x K = 1 / G syncs (batches) per epoch, both layouts, both rates
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_tf_golden.py -m gpu -q -p no:cacheprovider -k "simgcl or sept" > $O/r04_pytest_b.log 2>&1; echo "pytest exit $?"; tail -3 $O/r04_pytest_b.log | cut -c1-200
python - > /tmp/plan6.json <<'PY'
import json
c=[]
Y="yelp2018-clustered"
for lr0,ep in ((0.01,40),(0.05,20)):
    for world in (2,4,8):
        for K in (1,world):
            for layout in ("replicated","sharded"):
                if layout=="sharded" and K==1 and world==2: continue      # 2 ranks sharded has 2 batches anyway
                c.append(dict(dataset=Y,lr0=lr0,seed=7,mode="item",epochs=ep,eval_every=5,world=world,layout=layout,syncs=K))
print(json.dumps(c))
PY
timeout 1500 python tools/paired_recall.py $O/r04_paired_plan6.json /tmp/plan6.json > $O/r04_paired_plan6.log 2>&1; echo "plan6 exit $?"; grep -v "^{" $O/r04_paired_plan6.log | tail -5; grep "^{" $O/r04_paired_plan6.log | cut -c1-330
