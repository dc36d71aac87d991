#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/probe_hbm_resident_schedules.py > $O/r03_hbm_resident_schedules.log 2>&1; echo "exit $?"; tail -5 $O/r03_hbm_resident_schedules.log | cut -c1-400
