#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bpr.py -m gpu -q -s -p no:cacheprovider -k "deferred" > $O/r03_deferred_tests.log 2>&1; echo "deferred tests exit $?"; grep -E "passed|failed|deferred lr0|full grid|Error|assert " $O/r03_deferred_tests.log | cut -c1-260 | tail -20
for sched in item item-deferred; do
  timeout 200 python bench.py --schedule $sched --no-cpu-baseline > $O/r03_bench_$sched.json 2> $O/r03_bench_$sched.err
  python - <<PY
import json
d = json.load(open("$O/r03_bench_$sched.json"))
print("$sched", "value", round(d["value"] / 1e9, 4), "G/s  ms/epoch", round(d["config"]["ms_per_epoch"], 4), "kernel ms", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 4), "recall", d.get("recall_at_20"))
PY
done
