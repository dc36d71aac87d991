#!/bin/bash
# gpurun helper: bench + rocprofv3 kernel stats + HBM counters (separate pmc passes).  usage: tools/run_bench_prof.sh [round-tag]
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench exit $?"; cat $O/bench_$TAG.json
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $TAG -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_stats.log 2>&1; echo "stats exit $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o $TAG -- python $R/bench.py --steps 2 --warmup 1 --epochs-per-step 4 --no-cpu-baseline --no-extras > $O/prof_fetch.log 2>&1; echo "fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o $TAG -- python $R/bench.py --steps 2 --warmup 1 --epochs-per-step 4 --no-cpu-baseline --no-extras > $O/prof_write.log 2>&1; echo "write exit $?"
cd $R && python tools/summarize_prof.py $TAG > $O/summarize_$TAG.log 2>&1; echo "summarize exit $?"; head -12 $O/summarize_$TAG.log
