#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_bpr.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider -k "deferred or sharded or rccl_world_one or bench" > $O/r03_t25.log 2>&1; echo "tests exit $?"; tail -3 $O/r03_t25.log | cut -c1-300
QREC_FORCE_DIST=1 MASTER_PORT=29611 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras > $O/r03_shard_final.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r03_shard_final.json')); print('sharded world1', d['config']['ms_per_epoch'], d['config']['plan'])"
