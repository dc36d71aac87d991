#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
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
for rep in 1 2; do
run inside_nopipe "--no-shard-pipeline"
run start_nopipe "--no-shard-pipeline --no-plan-inside"
run inside_piped ""
run inside_nopipe_2b "--no-shard-pipeline --shard-batch 700000"
run start_nopipe_2b "--no-shard-pipeline --no-plan-inside --shard-batch 700000"
done
timeout 100 python bench.py --no-cpu-baseline --no-extras > $O/r03_plain.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r03_plain.json')); print('plain', d['config']['ms_per_epoch'])"
