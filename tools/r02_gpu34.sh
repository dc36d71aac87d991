#!/bin/bash
# the HIP trainers against the reference-run fixtures, one process per test, full logs kept
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for t in lightgcn bpr_tf ngcf simgcl; do
  timeout 100 python -X faulthandler -m pytest tests/test_gpu_tf_golden.py -q -x -k "test_${t}_trainer" > $O/tfg_$t.log 2>&1; rc=$?
  echo "$t exit $rc: $(tail -1 $O/tfg_$t.log | cut -c1-150)"
  if [ $rc -ne 0 ]; then grep -n "Error\|error\|assert\|Fatal\|File \"/root/repo" $O/tfg_$t.log | head -12; fi
done
