#!/bin/bash
# gpurun helper: bench + rocprofv3 kernel stats + HBM counters (separate pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cat $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r01 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_stats.log 2>&1; echo "stats exit $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o r01 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_fetch.log 2>&1; echo "fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o r01 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_write.log 2>&1; echo "write exit $?"
find $O/prof_stats $O/prof_fetch $O/prof_write -type f | head -30
