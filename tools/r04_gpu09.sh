#!/bin/bash
# round 4, call 9: the whole GPU suite with the parity ledger on; then the bench line + rocprofv3 kernel stats + PMC passes (tools/run_bench_prof.sh r04)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/r04_parity.jsonl
QREC_PARITY_LOG=$O/r04_parity.jsonl timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $O/r04_pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -22 $O/r04_pytest_gpu.log | cut -c1-220
bash tools/run_bench_prof.sh r04 2>&1 | cut -c1-600 | tail -40
python tools/summarize_parity.py $O/r04_parity.jsonl profiles/r04_parity_errors.json | tail -30
cp profiles/r04_parity_errors.json profiles/r04_kernel_stats.txt profiles/r04_hbm_counters.json profiles/hbm_traffic.json $O/ 2>/dev/null
