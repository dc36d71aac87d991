#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/counters_avail.txt 2>&1; grep -c . $O/counters_avail.txt
grep -E "^\s*(Name|Counter_Name|Name:)" $O/counters_avail.txt | head -5
grep -o -E "SQ_[A-Z_0-9]+" $O/counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  REPS=1 timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$tag -o r02 -- python $R/tools/bench_eval.py child > $O/pmc_$tag.log 2>&1; echo "pmc [$c] exit $?"
  python - <<PY
import sqlite3,glob
for db in glob.glob("$O/pmc_$tag/*results.db"):
    con=sqlite3.connect(db)
    try:
        for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%score_filter2%' group by kernel_name, counter_name"):
            print("   %-32s n=%d avg=%.4g"%(r[1],r[2],r[3]))
    except Exception as e: print("ERR", e)
PY
done
