#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in yelp2018 yelp2018-clustered; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_lg_${shape}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_lg_${shape}_$ctr -o lg -- python $R/tools/bench_lightgcn.py --steps 20 --shape $shape > $O/pmc_lg.log 2>&1; echo "$shape $ctr exit $?"
  done
done
python - <<'P'
import sqlite3, json, glob
out={}
for shape in ("yelp2018","yelp2018-clustered"):
    out[shape]={}
    for ctr in ("FETCH_SIZE","WRITE_SIZE"):
        db=f"/root/repo/gpurun_out/pmc_lg_{shape}_{ctr}/lg_results.db"
        con=sqlite3.connect(db)
        for name,n,avg in con.execute(f"select kernel_name, count(*), avg(value) from counters_collection where counter_name='{ctr}' group by kernel_name"):
            k=name.split('(')[0][-40:]
            out[shape].setdefault(k,{})[ctr+"_KB_avg"]=avg; out[shape][k]["dispatches"]=n
json.dump(out,open('/root/repo/gpurun_out/lightgcn_pmc.json','w'),indent=1)
for shape,ks in out.items():
    for k,v in ks.items():
        if 'spmm_kernel' in k: print(shape,k,{a:round(b,1) if isinstance(b,float) else b for a,b in v.items()})
P
