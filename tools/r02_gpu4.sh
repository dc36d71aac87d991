#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -k "logical_ranks" > $O/t_dist.log 2>&1; echo "logical ranks exit $?"; tail -6 $O/t_dist.log
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', round(d['value']/1e9,3), 'G/s', round(d['config']['ms_per_epoch'],4), 'ms/epoch', round(d['roofline']['avg_launch_ms'],4), 'kernel ms', d['config']['epochs_per_step'])"; }
A="--no-cpu-baseline --no-extras"
QREC_FORCE_DIST=1 timeout 300 python bench.py $A > $O/f1.json 2>/dev/null; show $O/f1.json "force replicated (null stream)"
QREC_FORCE_DIST=1 QREC_BENCH_STREAM=1 timeout 300 python bench.py $A > $O/f2.json 2>/dev/null; show $O/f2.json "force replicated (explicit stream)"
QREC_FORCE_DIST=1 QREC_BENCH_NO_COMM=1 timeout 300 python bench.py $A > $O/f3.json 2>/dev/null; show $O/f3.json "force replicated (no communicator)"
QREC_BENCH_STREAM=1 timeout 300 python bench.py $A > $O/f4.json 2>/dev/null; show $O/f4.json "plain (explicit stream)"
timeout 300 python bench.py $A > $O/f5.json 2>/dev/null; show $O/f5.json "plain (null stream)"
mkdir -p $O/dump_a $O/dump_b
B="--steps 1 --warmup 0 --epochs-per-step 8 --no-cpu-baseline --no-extras --shape ml1m"
QREC_DIST_TEST_ONE_DEVICE=1 QREC_DIST_TEST_DUMP=$O/dump_a timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 $B --dist-mode sharded --shard-batch 100000 --schedule user > /dev/null 2>&1
QREC_DIST_TEST_ONE_DEVICE=1 QREC_DIST_TEST_DUMP=$O/dump_b timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 2 $B --dist-mode sharded --shard-batch 1000000 > /dev/null 2>&1
python - <<PY
import numpy as np
for d,t in (("dump_a","sharded 2 ranks weak, user-major"),("dump_b","sharded 2 ranks weak, item-major, ONE batch")):
    try:
        z=np.load("$O/%s/rank0.npz"%d); print(t, "loss", np.round(z["log"][:,0]).tolist(), "lr", np.round(z["log"][:,1],4).tolist())
    except Exception as e: print(d, "ERR", e)
PY
