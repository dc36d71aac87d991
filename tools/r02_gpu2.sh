#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bpr.py -x -q -m gpu -k "scheduled_exact or ordered" > $O/t_exact.log 2>&1; echo "exact tests exit $?"; tail -15 $O/t_exact.log
timeout 600 python tools/probe_exact.py > $O/probe_exact.json 2> $O/probe_exact.err; echo "probe exit $?"; tail -20 $O/probe_exact.json; tail -5 $O/probe_exact.err
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $O/t_dist.log 2>&1; echo "dist tests exit $?"; tail -15 $O/t_dist.log
for m in replicated sharded; do
  QREC_FORCE_DIST=1 timeout 300 python bench.py --dist-mode $m --no-cpu-baseline --no-extras > $O/bench_force_$m.json 2> $O/bench_force_$m.err; echo "force $m exit $?"; cat $O/bench_force_$m.json; tail -3 $O/bench_force_$m.err
done
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain exit $?"; cat $O/bench_plain.json
cd /tmp && export TMPDIR=/tmp
QREC_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_force -o r02 -- python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/prof_force.log 2>&1; echo "prof exit $?"
ls $O/prof_force | head
