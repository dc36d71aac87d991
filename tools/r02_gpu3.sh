#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for m in replicated sharded; do
  QREC_FORCE_DIST=1 timeout 300 python bench.py --dist-mode $m --no-cpu-baseline --no-extras > $O/bench_force_$m.json 2> $O/bench_force_$m.err; echo "force $m exit $?"; python -c "
import json,sys; d=json.load(open('$O/bench_force_$m.json')); print(d['value'], d['config']['ms_per_epoch'], d['config']['final_loss'], d['config']['epochs_per_step'])"; grep -i "librccl path" $O/bench_force_$m.err
done
# loss trajectories: plain vs sharded (world 1, real RCCL) vs 2 ranks on one device
mkdir -p $O/dump_plain $O/dump_sh1 $O/dump_sh2 $O/dump_rep2
A="--steps 1 --warmup 0 --epochs-per-step 8 --no-cpu-baseline --no-extras --shape ml1m"
QREC_DIST_TEST_DUMP=$O/dump_plain timeout 300 python bench.py $A > /dev/null 2>&1
QREC_FORCE_DIST=1 QREC_DIST_TEST_DUMP=$O/dump_sh1 timeout 300 python bench.py $A --dist-mode sharded --shard-batch 100000 > /dev/null 2>&1
QREC_DIST_TEST_ONE_DEVICE=1 QREC_DIST_TEST_DUMP=$O/dump_sh2 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 $A --dist-mode sharded --shard-batch 100000 > /dev/null 2>&1
QREC_DIST_TEST_ONE_DEVICE=1 QREC_DIST_TEST_DUMP=$O/dump_rep2 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 2 $A > /dev/null 2>&1
python - <<PY
import numpy as np
for d in ("dump_plain","dump_sh1","dump_sh2","dump_rep2"):
    try:
        z=np.load("$O/%s/rank0.npz"%d); print(d, "loss", np.round(z["log"][:,0]).tolist(), "lr", np.round(z["log"][:,1],4).tolist())
    except Exception as e: print(d, "ERR", e)
PY
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu > $O/t_dist.log 2>&1; echo "dist tests exit $?"; tail -8 $O/t_dist.log
