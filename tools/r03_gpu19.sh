#!/bin/bash
# round-3 validation: the whole GPU suite with the parity ledger, smoke, bench + rocprof passes, sharded world-1 table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_r03.jsonl
QREC_PARITY_LOG=$O/parity_r03.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_r03.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu_r03.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/run_bench_prof.sh r03 2>&1 | grep -E "exit|^ +[0-9]+ " | head -30
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
run default ""
run batch19 "--shard-batch 524288"
run batch21 "--shard-batch 2097152"
run nopi "--no-plan-inside"
run pipeline "--shard-pipeline"
run planstream "--plan-ahead"
QREC_FORCE_DIST=1 MASTER_PORT=29613 timeout 200 python bench.py --dist-mode replicated --no-cpu-baseline --no-extras > $O/r03_repl_world1.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r03_repl_world1.json')); print('replicated world1', d['config']['ms_per_epoch'])"
