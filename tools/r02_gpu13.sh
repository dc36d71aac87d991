#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for dbg in 0 4 6 1 3; do QREC_DBG=$dbg python tools/probe_ngcf_dense.py; done
