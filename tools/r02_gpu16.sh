#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/class_epochs.jsonl
for mode in exact throughput; do for E in 2 6; do
QREC_MODE=$mode QREC_BENCH_EPOCHS=$E python tools/bench_class_epoch.py LightGCN SimGCL NGCF >> $O/class_epochs.jsonl 2>$O/class.err || tail -3 $O/class.err
done; done
python - <<'P'
import json
rows=[json.loads(l) for l in open('/root/repo/gpurun_out/class_epochs.jsonl')]
out={}
for mode in ("exact","throughput"):
    a=[r for r in rows if next(iter(r.values()))["mode"]==mode]
    lo=[r for r in a if next(iter(r.values()))["epochs"]==2][0]; hi=[r for r in a if next(iter(r.values()))["epochs"]==6][0]
    out[mode]={m: {"steady_s_per_epoch": round((hi[m]["train_s"]-lo[m]["train_s"])/4,4), "first_two_epochs_s": lo[m]["train_s"], "six_epochs_s": hi[m]["train_s"], "eval_s_by_epoch": hi[m]["eval_s_by_epoch"]} for m in lo}
print(json.dumps(out))
json.dump(out, open('/root/repo/gpurun_out/class_epoch_steady.json','w'), indent=1)
P
