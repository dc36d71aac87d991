#!/bin/bash
# round 4, call 4: item-major stored order in runs of 0 (whole items) / 16 / 8 -- Recall gap and time; the auto regime (6 M triplets); the new bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for run in 0 16 8; do
  QREC_ITEM_RUN=$run timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>$O/r04_bench_run$run.err \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('item_run=$run', 'ms/epoch', round(d['config']['ms_per_epoch'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'G/s', round(d['value']/1e9,3))"
done 2>&1 | tee $O/r04_item_run_timing.txt
tail -3 $O/r04_bench_run8.err
python - > /tmp/plan4.json <<'PY'
import json
c=[]
Y="yelp2018-clustered"
for ds,runs,ev in ((Y,((0.05,20),(0.01,40)),5),("lastfm",((0.05,20),(0.01,40)),4)):
    for lr0,ep in runs:
        for run in (8,16,0):
            c.append(dict(dataset=ds,lr0=lr0,seed=7,mode="item",epochs=ep,eval_every=ev,own_order=True,item_run=run))
X="xl6m-clustered"
for mode in ("item","item-deferred:4","item-deferred"):
    c.append(dict(dataset=X,lr0=0.05,seed=7,mode=mode,epochs=18,eval_every=3))
for mode in ("item","item-deferred:4"):
    c.append(dict(dataset=X,lr0=0.01,seed=7,mode=mode,epochs=40,eval_every=5))
print(json.dumps(c))
PY
timeout 1500 python tools/paired_recall.py $O/r04_paired_plan4.json /tmp/plan4.json > $O/r04_paired_plan4.log 2>&1; echo "plan4 exit $?"; grep -v "^{" $O/r04_paired_plan4.log | tail -5; grep "^{" $O/r04_paired_plan4.log | cut -c1-330
timeout 600 python bench.py > $O/r04_bench_try1.json 2> $O/r04_bench_try1.err; echo "bench exit $?"; tail -3 $O/r04_bench_try1.err; python -c "
import json; d=json.load(open('$O/r04_bench_try1.json')); print({k:(v if not isinstance(v,(dict,list)) else '...') for k,v in d.items()}); print(json.dumps(d['recall_at_20'])[:1500]); print(json.dumps(d.get('other_configs'))[:2500]); print(json.dumps(d.get('roofline_hbm_resident'))[:600])"
