#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
SECONDS=0; timeout 600 python bench.py > $O/bench_r03.json 2> $O/bench_r03.err; echo "bench exit $? wall ${SECONDS}s"
python - <<PY
import json
d = json.load(open("$O/bench_r03.json"))
print("value", round(d["value"] / 1e9, 4), "frac", round(d["roofline"]["frac"], 4), "hbm", round(d["roofline_hbm_resident"]["frac"], 4))
x = d["deferred_negatives"]; print("deferred leg", round(x["ms_per_epoch"], 4), round(x["roofline_frac"], 3), "hbm-resident:", x["roofline_hbm_resident"]["workload"], round(x["roofline_hbm_resident"]["frac"], 4), round(x["roofline_hbm_resident"]["avg_launch_ms"], 2))
PY
