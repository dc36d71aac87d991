#!/bin/bash
# round 3, call 1: the whole GPU suite with the parity ledger on (every comparison's observed error), the six reference-run tests
# that had never executed included; then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_r03.jsonl
QREC_PARITY_LOG=$O/parity_r03.jsonl timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r03_pytest1.log 2>&1
echo "pytest exit $?"; tail -40 $O/r03_pytest1.log | cut -c1-260
timeout 300 python bench.py > $O/r03_bench1.json 2> $O/r03_bench1.err; echo "bench exit $?"; cut -c1-600 $O/r03_bench1.json
