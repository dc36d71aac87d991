#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f /tmp/cpu_long.npz
timeout 600 python tools/probe_deferred_long.py item /tmp/cpu_long.npz > $O/r03_long_item.json 2> $O/r03_long_item.err; echo "item exit $?"
QREC_DEFERRED_FRESH=0 timeout 300 python tools/probe_deferred_long.py item-deferred /tmp/cpu_long.npz > $O/r03_long_def0.json 2> $O/r03_long_def0.err; echo "def0 exit $?"
QREC_DEFERRED_FRESH=1 timeout 300 python tools/probe_deferred_long.py item-deferred /tmp/cpu_long.npz > $O/r03_long_def1.json 2> $O/r03_long_def1.err; echo "def1 exit $?"
python - <<PY
import json
for f in ("item", "def0", "def1"):
    try:
        d = json.loads(open("$O/r03_long_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["schedule"], d["fresh"], {k: (round(v["recall_gpu"], 5), round(v["recall_exact_order"], 5), round(v["abs_diff"], 5), round(v["loss_gpu"]), round(v["loss_exact_order"]), round(v["lr_gpu"], 4), round(v["lr_exact_order"], 4)) for k, v in d.items() if k.startswith("epochs")})
    except Exception as e:
        print(f, "failed", e, open("$O/r03_long_%s.err" % f).read()[-800:])
PY
