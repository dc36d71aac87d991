#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bpr.py tests/test_gpu_tf_golden.py -m gpu -q -p no:cacheprovider -k "sigmoid_underflows or sept_trainer or scheduled_exact or ordered" > $O/r03_t20.log 2>&1; echo "tests exit $?"; tail -3 $O/r03_t20.log | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline > $O/r03_b20.json 2> $O/r03_b20.err
python - <<PY
import json
d = json.load(open("$O/r03_b20.json"))
x = d["deferred_negatives"]
print("main", d["config"]["ms_per_epoch"], "leg", x["ms_per_epoch"], "kernels", x["avg_launch_ms"], "enqueue", x["host_enqueue_ms_per_epoch"])
PY
timeout 200 python bench.py --schedule item-deferred --no-cpu-baseline --no-extras > $O/r03_b20d.json 2> $O/r03_b20d.err
python - <<PY
import json
d = json.load(open("$O/r03_b20d.json"))
print("item-deferred main loop", d["config"]["ms_per_epoch"], d["roofline"]["avg_launch_ms"], "epochs/step", d["config"]["epochs_per_step"])
PY
