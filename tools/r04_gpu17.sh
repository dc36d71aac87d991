#!/bin/bash
# round 4: what the N > 1 bench line looks like -- two real processes on the ONE device of the box over the staged gloo transport (a functional run:
# the line marks itself INVALID_AS_BENCH), default shape, default flags as the driver types them; plus K = 2G reconciliations at 5x the rate
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
QREC_DIST_TEST_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/r04_bench_two_ranks_one_device.json 2> $O/r04_bench2.err; echo "bench --gpus 2 exit $?"; tail -3 $O/r04_bench2.err | cut -c1-300
python -c "
import json; d=json.load(open('$O/r04_bench_two_ranks_one_device.json'))
print({k:(v if not isinstance(v,(dict,list)) else '...') for k,v in d.items()})
print(d['config']['workload']); print(json.dumps(d['multi_gpu'])[:900]); print(json.dumps(d.get('weak_scaling'))[:500]); print(json.dumps(d.get('recall_at_20'))[:900])"
cat > /tmp/plan17.json <<'PLAN'
[{"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":4,"layout":"replicated","syncs":8},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":7,"mode":"item","epochs":20,"eval_every":5,"world":4,"layout":"replicated","syncs":16},
 {"dataset":"yelp2018-clustered","lr0":0.05,"seed":11,"mode":"item","epochs":20,"eval_every":5,"world":4,"layout":"replicated"}]
PLAN
timeout 600 python tools/paired_recall.py $O/r04_paired_plan17.json /tmp/plan17.json > $O/r04_paired_plan17.log 2>&1; echo "plan17 exit $?"; grep "^{" $O/r04_paired_plan17.log | cut -c1-300
