#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/probe_hbm_resident_groups.py > $O/r03_hbm_groups.log 2>&1; echo "exit $?"; tail -11 $O/r03_hbm_groups.log | cut -c1-200
