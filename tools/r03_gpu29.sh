#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_bpr.py -m gpu -q -s -p no:cacheprovider -k "deferred" > $O/r03_sub_tests.log 2>&1; echo "deferred tests exit $?"; grep -E "passed|failed|Error" $O/r03_sub_tests.log | cut -c1-200 | tail -4
for S in 1 4 8; do
QREC_DEFERRED_SUB=$S QREC_DEFERRED_SUB_CHUNK=32 timeout 300 python - <<PY 2>&1 | tail -1
import sys; sys.path.insert(0, "$R")
import bench as B
from qrec_amd import capi
capi.init(0)
r = B.hbm_resident_roofline(capi, schedule="item-deferred")
print("hbm-resident item-deferred S=$S:", round(r["avg_launch_ms"], 3), "ms", round(r["frac"], 4))
PY
done
