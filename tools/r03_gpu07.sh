#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider -k "sharded or rccl_binding or shard_plan or typed_with" > $O/r03_dist_tests.log 2>&1; echo "dist tests exit $?"; tail -6 $O/r03_dist_tests.log | cut -c1-250
for mode in "" "--no-shard-pipeline"; do
  QREC_FORCE_DIST=1 MASTER_PORT=29549 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras $mode > $O/r03_force_dist_sharded${mode}.json 2> $O/r03_force_dist_sharded${mode}.err
  echo "sharded world1 $mode: $(python -c "import json; d=json.load(open('$O/r03_force_dist_sharded${mode}.json')); print('ms/epoch', round(d['config']['ms_per_epoch'],4), 'batches', d['config']['batches_per_epoch'], 'piped', d['config']['fetch_pipelined'], 'loss', round(d['config']['final_loss'],1))" 2>&1 | tail -1)"
done
QREC_FORCE_DIST=1 MASTER_PORT=29549 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras --shard-batch 262144 > $O/r03_force_dist_sharded_b18.json 2>/dev/null; python -c "import json; d=json.load(open('$O/r03_force_dist_sharded_b18.json')); print('shard-batch 2^18: ms/epoch', round(d['config']['ms_per_epoch'],4), 'batches', d['config']['batches_per_epoch'])"
QREC_FORCE_DIST=1 MASTER_PORT=29549 timeout 200 python bench.py --dist-mode replicated --no-cpu-baseline --no-extras > $O/r03_force_dist_replicated.json 2>/dev/null; python -c "import json; d=json.load(open('$O/r03_force_dist_replicated.json')); print('replicated world1: ms/epoch', round(d['config']['ms_per_epoch'],4))"
timeout 200 python bench.py --no-cpu-baseline --no-extras > $O/r03_plain.json 2>/dev/null; python -c "import json; d=json.load(open('$O/r03_plain.json')); print('plain: ms/epoch', round(d['config']['ms_per_epoch'],4))"
timeout 120 tools/ubench/atomics4 > $O/r03_ubench_atomics4.txt 2>&1; echo "ubench exit $?"; head -8 $O/r03_ubench_atomics4.txt
