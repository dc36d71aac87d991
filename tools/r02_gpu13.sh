#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "--- occ1"; QREC_DBG_CLK=1 python tools/probe_ngcf_dense.py 2>&1 | tail -14
echo "--- occ2"; QREC_NGCF_OCC2=1 QREC_DBG_CLK=1 python tools/probe_ngcf_dense.py 2>&1 | tail -14
