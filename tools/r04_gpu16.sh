#!/bin/bash
# round 4: the whole GPU suite once more after engine.launch_chunk (the driver's command line), ledger refreshed
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/r04_parity.jsonl
QREC_PARITY_LOG=$O/r04_parity.jsonl timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r04_pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -6 $O/r04_pytest_gpu.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/summarize_parity.py $O/r04_parity.jsonl $O/r04_parity_errors.json | head -3
