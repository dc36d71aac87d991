#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for fresh in 1 0; do
QREC_DEFERRED_FRESH=$fresh timeout 900 python -m pytest tests/test_gpu_bpr.py -m gpu -q -s -p no:cacheprovider -k "deferred" > $O/r03_deferred_fresh$fresh.log 2>&1; echo "fresh=$fresh tests exit $?"; grep -E "passed|failed|deferred lr0|full grid|observed" $O/r03_deferred_fresh$fresh.log | cut -c1-260 | tail -8
done
QREC_DEFERRED_FRESH=1 timeout 200 python bench.py --schedule item-deferred --no-cpu-baseline > $O/r03_bench_fresh.json 2> $O/r03_bench_fresh.err
python - <<PY
import json
d = json.load(open("$O/r03_bench_fresh.json"))
print("fresh: value", round(d["value"] / 1e9, 4), "ms/epoch", round(d["config"]["ms_per_epoch"], 4), "kernel ms", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 4), "recall", d.get("recall_at_20"))
PY
