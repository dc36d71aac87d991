#!/bin/bash
# the whole GPU suite twice with the parity ledger (the spread of the nondeterministic comparisons), smoke, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_r03.jsonl
for rep in 1 2; do
QREC_PARITY_LOG=$O/parity_r03.jsonl timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_r03_$rep.log 2>&1; echo "pytest $rep exit $?"; tail -3 $O/pytest_gpu_r03_$rep.log | cut -c1-300
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench_r03.json 2> $O/bench_r03.err; echo "bench exit $?"
python - <<PY
import json
d = json.load(open("$O/bench_r03.json"))
print("value", round(d["value"] / 1e9, 4), "ms/epoch", round(d["config"]["ms_per_epoch"], 4), "frac", round(d["roofline"]["frac"], 4))
x = d["deferred_negatives"]; print("deferred leg", round(x["ms_per_epoch"], 4), round(x["avg_launch_ms"], 4), round(x["roofline_frac"], 3), x["recall_at_20"]["abs_diff"])
print("exact", d["exact_mode"]["value"], "recall", d["recall_at_20"]["abs_diff"], "cpu", d["cpu_baseline"]["value"])
PY
QREC_FORCE_DIST=1 MASTER_PORT=29611 timeout 200 python bench.py --dist-mode sharded --no-cpu-baseline --no-extras > $O/r03_shard_final.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r03_shard_final.json')); print('sharded world1', d['config']['ms_per_epoch'], d['config']['plan'])"
