#!/bin/bash
# round 4, call 1: paired Recall@20 harness (quick plan first, then a focused plan) + deferred sub-epoch chunk sweep (timing)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python tools/paired_recall.py $O/r04_paired_quick.json quick > $O/r04_paired_quick.log 2>&1; echo "quick exit $?"; tail -6 $O/r04_paired_quick.log | cut -c1-400
# timing sweep of the deferred schedule: S sub-epochs, pass-A chunk sized so that a range is ~1 / ~2 rounds of the 16,384-group grid
for cfg in "1 34" "2 38" "2 19" "3 26" "4 19" "4 20" "4 10" "6 13" "8 10"; do
  set -- $cfg
  QREC_DEFERRED_SUB=$1 QREC_DEFERRED_SUB_CHUNK=$2 timeout 200 python bench.py --schedule item-deferred --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S=$1 chunk=$2', 'ms/epoch', round(d['config']['ms_per_epoch'],4), 'kernels ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))"
done 2>&1 | tee $O/r04_deferred_chunk_sweep.txt
cat > /tmp/plan1.json <<'PLAN'
[{"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred:4","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred:4:19","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred:2:38","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred:3:26","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item-deferred:8:10","epochs":20,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item","epochs":40,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item-deferred","epochs":40,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.01,"seed":7,"mode":"item-deferred:4:19","epochs":40,"eval_every":5},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":2,"layout":"replicated"},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":4,"layout":"replicated"},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":2,"layout":"sharded"},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":4,"layout":"sharded"},
 {"dataset":"lastfm","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":2},
 {"dataset":"lastfm","lr0":0.05,"seed":7,"mode":"item-deferred","epochs":20,"eval_every":2},
 {"dataset":"lastfm","lr0":0.01,"seed":7,"mode":"item","epochs":40,"eval_every":4},
 {"dataset":"lastfm","lr0":0.01,"seed":7,"mode":"item-deferred","epochs":40,"eval_every":4}]
PLAN
timeout 900 python tools/paired_recall.py $O/r04_paired_plan1.json /tmp/plan1.json > $O/r04_paired_plan1.log 2>&1; echo "plan1 exit $?"; tail -20 $O/r04_paired_plan1.log | cut -c1-300
