#!/bin/bash
# gpurun helper for a re-check after a failed `tests` step of gpu_steps.sh (pytest -x stops at the first failure): the named test first, then
# the tests the stopped run never reached (a file of node ids), then the bench line.  Everything under its own `timeout`.
# usage: gpurun --timeout S -- 'bash tools/gpu_recheck.sh <tag> <first-test-node-id> <file-of-node-ids> [-k expr of quick extra tests]'
TAG=$1; FIRST=$2; REST=$3; EXTRA=$4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export QREC_PARITY_LOG=$O/${TAG}_parity_raw.jsonl
T0=$(date +%s)
if [ -n "$EXTRA" ]; then
  timeout 200 python -m pytest tests -m gpu -q -k "$EXTRA" > $O/${TAG}_extra.log 2>&1; echo "extra exit $? [$(( $(date +%s) - T0 )) s]"; tail -2 $O/${TAG}_extra.log | cut -c1-300
fi
timeout 260 python -m pytest "$FIRST" -q > $O/${TAG}_first.log 2>&1; echo "first exit $? [$(( $(date +%s) - T0 )) s]"; tail -12 $O/${TAG}_first.log | cut -c1-500
timeout 420 python -m pytest @$REST -q > $O/${TAG}_rest.log 2>&1; echo "rest exit $? [$(( $(date +%s) - T0 )) s]"; tail -6 $O/${TAG}_rest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 150 python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench exit $? [$(( $(date +%s) - T0 )) s]"; head -c 600 $O/${TAG}_bench.json; echo
