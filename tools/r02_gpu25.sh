#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_mfma_ngcf $O/pmc_mfma_eval $O/pmc_mfma_simgcl
rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/pmc_mfma_ngcf -o m -- python $R/tools/prof_ngcf.py > $O/pmc_mfma.log 2>&1; echo "ngcf exit $?"
rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/pmc_mfma_eval -o m -- python $R/tools/bench_eval.py child > $O/pmc_mfma2.log 2>&1; echo "eval exit $?"
rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/pmc_mfma_simgcl -o m -- python $R/tools/bench_eval_simgcl.py --skip-eval > $O/pmc_mfma3.log 2>&1; echo "simgcl exit $?"
python - <<'P'
import sqlite3, json
out={}
for tag in ("ngcf","eval","simgcl"):
    con=sqlite3.connect(f"/root/repo/gpurun_out/pmc_mfma_{tag}/m_results.db")
    for name,n,avg,mx in con.execute("select kernel_name, count(*), avg(value), max(value) from counters_collection where counter_name='MfmaUtil' group by kernel_name"):
        if avg > 0.5: out[name[:70]]={"dispatches":n,"mfma_util_pct_avg":round(avg,1),"mfma_util_pct_max":round(mx,1),"from":tag}
json.dump(out,open('/root/repo/gpurun_out/mfma_util.json','w'),indent=1)
for k,v in out.items(): print(v["mfma_util_pct_avg"], v["dispatches"], k)
P
