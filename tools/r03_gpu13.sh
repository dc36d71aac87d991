#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bpr.py tests/test_gpu_eval.py -m gpu -q -p no:cacheprovider -k "deferred or adversarial" > $O/r03_t13.log 2>&1; echo "tests exit $?"; grep -E "passed|failed|Error|assert " $O/r03_t13.log | cut -c1-260 | tail -20
timeout 400 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "bench exit $?"
python - <<PY
import json
d = json.load(open("$O/r03_bench_default.json"))
print("value", round(d["value"] / 1e9, 4), "ms/epoch", round(d["config"]["ms_per_epoch"], 4), "frac", round(d["roofline"]["frac"], 4), d["roofline"].get("atomic_unit_floor"))
print("recall", d["recall_at_20"]); print("exact", d["exact_mode"]["value"]); print("cpu", d["cpu_baseline"]["value"], d.get("vs_cpu_port"))
print("deferred", json.dumps(d.get("deferred_negatives")))
print("hbm", d["roofline_hbm_resident"]["frac"])
PY
