#!/bin/bash
# gpurun helper (round 2, first call): new multi-GPU pieces on the one device + the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $O/t_dist.log 2>&1; echo "dist tests exit $?"; tail -15 $O/t_dist.log
timeout 600 python bench.py > $O/bench_r02a.json 2> $O/bench_r02a.err; echo "bench exit $?"; cat $O/bench_r02a.json; tail -5 $O/bench_r02a.err
for m in replicated sharded; do
  QREC_FORCE_DIST=1 timeout 300 python bench.py --dist-mode $m --no-cpu-baseline --no-extras > $O/bench_force_$m.json 2> $O/bench_force_$m.err; echo "force $m exit $?"; cat $O/bench_force_$m.json; tail -3 $O/bench_force_$m.err
done
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_dist.py > $O/t_all.log 2>&1; echo "all gpu tests exit $?"; tail -8 $O/t_all.log
