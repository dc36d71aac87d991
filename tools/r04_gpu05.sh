#!/bin/bash
# round 4, call 5: the whole GPU suite with the parity ledger switched on
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/r04_parity_a.jsonl
QREC_PARITY_LOG=$O/r04_parity_a.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > $O/r04_pytest_gpu_a.log 2>&1; echo "pytest exit $?"
tail -45 $O/r04_pytest_gpu_a.log | cut -c1-250
