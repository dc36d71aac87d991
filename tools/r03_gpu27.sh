#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bpr.py -m gpu -q -s -p no:cacheprovider -k "deferred_sub_epochs" > $O/r03_sub_tests.log 2>&1; echo "sub tests exit $?"; grep -E "passed|failed|full grid|observed|Error" $O/r03_sub_tests.log | cut -c1-260 | tail -8
variant() { # S chunk fresh
  export QREC_DEFERRED_SUB=$1 QREC_DEFERRED_SUB_CHUNK=$2 QREC_DEFERRED_FRESH=$3
  timeout 200 python bench.py --schedule item-deferred --no-cpu-baseline --no-extras > $O/r03_sub_b.json 2> $O/r03_sub_b.err
  timeout 300 python -m pytest tests/test_gpu_bpr.py -m gpu -q -s -p no:cacheprovider -k "deferred_recall" > $O/r03_sub_r.log 2>&1
  python - <<PY
import json, re
try:
    d = json.load(open("$O/r03_sub_b.json")); t = "ms/epoch %.4f kernels %.4f" % (d["config"]["ms_per_epoch"], d["roofline"]["avg_launch_ms"])
except Exception as e:
    t = "bench failed " + open("$O/r03_sub_b.err").read()[-300:]
rec = re.findall(r"deferred lr0 (\S+) Recall@20 exact-order (\S+) deferred (\S+)", open("$O/r03_sub_r.log").read())
print("S=$1 chunk=$2 fresh=$3:", t, " recall gaps:", [(a, round(abs(float(b) - float(c)), 5)) for a, b, c in rec])
PY
}
variant 1 8 0
variant 4 8 0
variant 4 16 0
variant 2 16 0
variant 8 8 0
variant 4 8 1
