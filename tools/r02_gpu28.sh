#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/bench_configs.py > $O/other_configs.json 2>$O/cfg.err || tail -3 $O/cfg.err
python -c "
import json; d=json.load(open('$O/other_configs.json'))
for k,v in d.items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms','gtps','frac','ms_per_step')})"
