#!/bin/bash
# the deferred-negatives schedule as the bench's main line under rocprofv3: kernel stats, FETCH_SIZE / WRITE_SIZE in separate passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_def_stats $O/prof_def_fetch $O/prof_def_write
rocprofv3 --kernel-trace --stats -d $O/prof_def_stats -o r03def -- python $R/bench.py --schedule item-deferred --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_def_stats.log 2>&1; echo "stats exit $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_def_fetch -o r03def -- python $R/bench.py --schedule item-deferred --steps 2 --warmup 1 --epochs-per-step 4 --no-cpu-baseline --no-extras > $O/prof_def_fetch.log 2>&1; echo "fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_def_write -o r03def -- python $R/bench.py --schedule item-deferred --steps 2 --warmup 1 --epochs-per-step 4 --no-cpu-baseline --no-extras > $O/prof_def_write.log 2>&1; echo "write exit $?"
cd $R
timeout 300 python bench.py --no-extras > $O/bench_r03_cpu.json 2> $O/bench_r03_cpu.err; python -c "
import json; d=json.load(open('$O/bench_r03_cpu.json')); print(json.dumps(d['cpu_baseline'])[:900]); print(d.get('vs_cpu_port'), d.get('vs_reference_loop_here'))"
