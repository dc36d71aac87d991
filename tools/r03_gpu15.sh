#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -k "sharded or rccl_world_one" > $O/r03_t15.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|Error|assert " $O/r03_t15.log | cut -c1-260 | tail -12
run() { # name, flags
  QREC_FORCE_DIST=1 MASTER_PORT=29611 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras $2 > $O/r03_shard_$1.json 2> $O/r03_shard_$1.err
  python - <<PY
import json
try:
    d = json.load(open("$O/r03_shard_$1.json"))
    print("$1", "ms/epoch", round(d["config"]["ms_per_epoch"], 4), "batches", d["config"]["batches_per_epoch"], "piped", d["config"]["fetch_pipelined"], "ahead", d["config"]["plan_ahead"], "loss", round(d["config"]["final_loss"]))
except Exception as e:
    print("$1 failed", e); print(open("$O/r03_shard_$1.err").read()[-1500:])
PY
}
run base "--no-shard-pipeline --no-plan-ahead"
run ahead "--no-shard-pipeline"
run piped "--no-plan-ahead"
run both ""
run ahead_1batch "--no-shard-pipeline --shard-batch 2097152"
run base_1batch "--no-shard-pipeline --no-plan-ahead --shard-batch 2097152"
run ahead_2batch "--no-shard-pipeline --shard-batch 700000"
run both_2batch "--shard-batch 700000"
