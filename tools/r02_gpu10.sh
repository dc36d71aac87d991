#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "ALL gpu tests exit $?"; tail -6 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
bash tools/run_bench_prof.sh r02
mkdir -p $O/profiles_out && cp $R/profiles/r02_* $R/profiles/hbm_traffic.json $O/profiles_out/ 2>/dev/null
QREC_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_force_replicated.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_force_replicated.json')); print('force replicated', d['config']['ms_per_epoch'], d['value'])"
QREC_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras --dist-mode sharded > $O/bench_force_sharded.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_force_sharded.json')); print('force sharded', d['config']['ms_per_epoch'], d['value'], d['config']['batches_per_epoch'])"
