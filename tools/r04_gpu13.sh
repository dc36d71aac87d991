#!/bin/bash
# round 4, final: the whole GPU suite as the driver types it (-x) with the parity ledger on, smoke(), then the bench line + rocprofv3 stats + PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/r04_parity.jsonl
QREC_PARITY_LOG=$O/r04_parity.jsonl timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r04_pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -6 $O/r04_pytest_gpu.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/run_bench_prof.sh r04 2>&1 | cut -c1-400 | tail -16
python tools/summarize_parity.py $O/r04_parity.jsonl profiles/r04_parity_errors.json | tail -12
cp profiles/r04_parity_errors.json profiles/r04_kernel_stats.txt profiles/r04_hbm_counters.json profiles/hbm_traffic.json $O/ 2>/dev/null
