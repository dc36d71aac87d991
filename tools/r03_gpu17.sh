#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -k "sharded or rccl_world_one" > $O/r03_t17.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|Error|assert " $O/r03_t17.log | cut -c1-260 | tail -12
run() { # name, flags
  QREC_FORCE_DIST=1 MASTER_PORT=29611 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras $2 > $O/r03_shard_$1.json 2> $O/r03_shard_$1.err
  python - <<PY
import json
try:
    d = json.load(open("$O/r03_shard_$1.json"))
    print("$1", "ms/epoch", round(d["config"]["ms_per_epoch"], 4), "batches", d["config"]["batches_per_epoch"], "piped", d["config"]["fetch_pipelined"], "plan:", d["config"]["plan"], "loss", round(d["config"]["final_loss"]))
except Exception as e:
    print("$1 failed", e); print(open("$O/r03_shard_$1.err").read()[-1500:])
PY
}
run inside_nopipe "--no-shard-pipeline"
run start_nopipe "--no-shard-pipeline --no-plan-inside"
run inside_piped ""
run start_piped "--no-plan-inside"
run inside_nopipe_2b "--no-shard-pipeline --shard-batch 700000"
run start_nopipe_2b "--no-shard-pipeline --no-plan-inside --shard-batch 700000"
run inside_piped_2b "--shard-batch 700000"
run start_nopipe_1b "--no-shard-pipeline --no-plan-inside --shard-batch 2097152"
