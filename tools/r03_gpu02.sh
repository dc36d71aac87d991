#!/bin/bash
# round 3, call 2: the four-triplets-per-wavefront exact kernel (tests + probe), the NGCF row-partition discrepancy by step,
# bench.py typed with --gpus 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bpr.py -m gpu -q -x -p no:cacheprovider -k "ordered or scheduled or exact or bpr_model_end_to_end or pipelined or basicmf or pmf or svdpp or tbpr or mf_family" > $O/r03_exact_tests.log 2>&1
echo "exact tests exit $?"; tail -15 $O/r03_exact_tests.log | cut -c1-220
timeout 300 python tools/probe_exact.py > $O/r03_exact_probe.log 2>&1; echo "probe exit $?"; grep -v "^{" $O/r03_exact_probe.log | cut -c1-250 | tail -20
timeout 300 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider -k "typed_with_gpus_2" > $O/r03_selflaunch.log 2>&1; echo "selflaunch exit $?"; tail -5 $O/r03_selflaunch.log | cut -c1-300
# NGCF, row-partitioned class on two ranks vs one rank: where do the losses part?
export QREC_SEED=11 QREC_DIST_TEST_ONE_DEVICE=1 QREC_GRAPH_DIST=rows
mkdir -p $O/ngcf_one $O/ngcf_two
timeout 200 python tests/graph_dp_worker.py NGCF 1024 $O/ngcf_one > $O/ngcf_one.log 2>&1; echo "one exit $?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 tests/graph_dp_worker.py NGCF 1024 $O/ngcf_two > $O/ngcf_two.log 2>&1; echo "two exit $?"
python - <<'PY'
import numpy as np, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
a = np.load(O + "/ngcf_one/rank0.npz"); b = np.load(O + "/ngcf_two/rank0.npz")
la, lb = a["losses"], b["losses"]
print("steps", la.size, lb.size)
print("rel diff by step:", " ".join(f"{abs(x - y) / abs(x):.1e}" for x, y in zip(la, lb)))
PY
