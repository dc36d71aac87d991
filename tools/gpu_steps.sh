#!/bin/bash
# The ONE gpurun helper (round 5; replaces the per-call tools/r0N_gpuNN.sh scripts of rounds 2-4, which stay in git history).
# usage:  gpurun --timeout S -- 'bash tools/gpu_steps.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/ (merged back by gpurun); summaries to be judged are copied into profiles/ by hand afterwards.
#
# steps
#   tests[:expr]         pytest -m gpu (optionally -k expr), parity ledger into gpurun_out/<tag>_parity_raw.jsonl
#   tests-all[:expr]     the same without -x (every failure of the selection in one call), ledger into <tag>_parity_all_raw.jsonl
#   bench                python bench.py (default flags) -> <tag>_bench.json
#   bench-prof           bench + rocprofv3 --kernel-trace --stats + the two HBM counter passes (separate) + tools/summarize_prof.py
#   recall:<plan>        tools/paired_recall.py <tag>_recall_<plan>.json <plan>       (plan: bpr-conf | full | quick | a JSON file under tools/plans/)
#   spmm-counters        FETCH_SIZE / WRITE_SIZE (separate passes) of the forward SpMM on the structureless and the planted-community graph
#                        -> <tag>_lightgcn_hbm_counters.json (bytes past the XCD L2s per launch, over-fetch vs algorithmic bytes)
#   mfma-util            MfmaUtil per MFMA kernel (NGCF step, evaluation, SimGCL step) -> <tag>_mfma_util.json
#   pmc:<m>:<C1,C2,..>   one --pmc pass per counter over one config's step -> <tag>_<m>_counters.json
#   stats:<m>            rocprofv3 --kernel-trace --stats of one config's step (m: lightgcn | simgcl | ngcf | eval) -> <tag>_<m>_kernel_stats.txt
#   bpr-counters         the item-major BPR kernel under rocprofv3: kernel stats + one --pmc pass PER counter (atomic / request / busy / wait
#                        counters that `rocprofv3 -L` lists on this box; FETCH_SIZE, WRITE_SIZE) at the Yelp2018 shape and on the HBM-resident
#                        slice, atomic and load+store P[u] -> <tag>_bpr_counters.json, <tag>_bpr_<w>_<p>_kernel_stats.txt, <tag>_counters_available.txt
#   py:<script>[:args]   python tools/<script>.py args... (args separated by ':'), stdout -> <tag>_<script>.log
#   profpy:<script>[:args]  the same under rocprofv3 --kernel-trace --stats -> <tag>_<script>_kernel_stats.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp

step_cmd() {
  case $1 in
    lightgcn) echo "python $R/tools/bench_lightgcn.py --steps 60";;
    simgcl) echo "python $R/tools/bench_eval_simgcl.py --skip-eval";;
    ngcf) echo "python $R/tools/prof_ngcf.py";;
    eval) echo "python $R/tools/bench_eval.py child";;
  esac
}

for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [[ "$STEP" == *:* ]] && ARG=${STEP#*:}
  T0=$(date +%s)
  case $KIND in
    tests)
      cd $R
      QREC_PARITY_LOG=$O/${TAG}_parity_raw.jsonl timeout 3000 python -m pytest tests -m gpu -x -q ${ARG:+-k "$ARG"} > $O/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -5 $O/${TAG}_tests.log
      cd /tmp;;
    tests-all)
      cd $R
      QREC_PARITY_LOG=$O/${TAG}_parity_all_raw.jsonl timeout 3000 python -m pytest tests -m gpu -q ${ARG:+-k "$ARG"} > $O/${TAG}_tests_all.log 2>&1; echo "tests-all exit $?"
      grep -E "^(FAILED|ERROR)|passed|failed" $O/${TAG}_tests_all.log | cut -c1-300 | tail -40
      cd /tmp;;
    bench)
      timeout 900 python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench exit $?"; head -c 1500 $O/${TAG}_bench.json; echo;;
    bench-prof)
      bash $R/tools/run_bench_prof.sh $TAG;;
    recall)
      PLAN=$ARG; [ -f $R/tools/plans/$ARG.json ] && PLAN=$R/tools/plans/$ARG.json
      cd $R; timeout 3000 python tools/paired_recall.py $O/${TAG}_recall_${ARG}.json $PLAN > $O/${TAG}_recall_${ARG}.log 2>&1; echo "recall $ARG exit $?"
      grep "over seeds" $O/${TAG}_recall_${ARG}.log | cut -c1-900; tail -3 $O/${TAG}_recall_${ARG}.log | cut -c1-400; cd /tmp;;
    spmm-counters)
      SPECS=""
      for SH in yelp2018 yelp2018-clustered; do
        for ctr in FETCH_SIZE WRITE_SIZE; do
          rm -rf $O/pmc_lg_${SH}_$ctr
          rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_lg_${SH}_$ctr -o lg -- python $R/tools/bench_lightgcn.py --steps 20 --shape $SH > $O/pmc_lg_${SH}_$ctr.log 2>&1; echo "$SH $ctr exit $?"
        done
        SPECS="$SPECS $SH=$O/pmc_lg_${SH}_FETCH_SIZE,$O/pmc_lg_${SH}_WRITE_SIZE"
      done
      python $R/tools/summarize_spmm_counters.py $O/${TAG}_lightgcn_hbm_counters.json $SPECS
      rm -rf $O/pmc_lg_*_FETCH_SIZE $O/pmc_lg_*_WRITE_SIZE;;
    mfma-util)
      for m in ngcf eval simgcl; do
        rm -rf $O/pmc_mfma_$m
        rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/pmc_mfma_$m -o m -- $(step_cmd $m) > $O/pmc_mfma_$m.log 2>&1; echo "$m exit $?"
      done
      python $R/tools/summarize_pmc.py $O/${TAG}_mfma_util.json MfmaUtil=$O/pmc_mfma_ngcf MfmaUtil=$O/pmc_mfma_eval MfmaUtil=$O/pmc_mfma_simgcl;;
    pmc)
      # pmc:<m>:<C1,C2,...>   one rocprofv3 --pmc pass PER counter over one config's step (m as in stats:) -> <tag>_<m>_counters.json
      M=${ARG%%:*}; CS=${ARG#*:}; SPECS=""
      for c in ${CS//,/ }; do
        rm -rf $O/pmc_${M}_$c
        timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${M}_$c -o k -- $(step_cmd $M) > $O/pmc_${M}_$c.log 2>&1 || echo "$c failed"
        SPECS="$SPECS $c=$O/pmc_${M}_$c"
      done
      python $R/tools/summarize_pmc.py $O/${TAG}_${M}_counters.json $SPECS | cut -c1-250 | head -40
      rm -rf $O/pmc_${M}_*;;
    stats)
      rm -rf $O/prof_$ARG
      rocprofv3 --kernel-trace --stats -d $O/prof_$ARG -o $ARG -- $(step_cmd $ARG) > $O/prof_$ARG.log 2>&1; echo "$ARG exit $?"
      db=$(ls $O/prof_$ARG/*_results.db $O/prof_$ARG/*/*_results.db 2>/dev/null | head -1)
      python $R/tools/summarize_stats.py $db $O/${TAG}_${ARG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $(step_cmd $ARG)   [$TAG, Yelp2018 shape d=64; profiled run]" | head -12;;
    bpr-counters)
      rocprofv3 -L > $O/${TAG}_counters_list_full.txt 2>&1
      grep -oE "\b(TCC|TCP|SQ|GRBM|TA|TD)_[A-Za-z0-9_]+" $O/${TAG}_counters_list_full.txt | sort -u > $O/${TAG}_counters_available.txt
      wc -l $O/${TAG}_counters_available.txt
      WANT="TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA_ATOMIC_sum TCC_REQ_sum TCC_BUSY_sum TCC_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_sum TCC_TAG_STALL_sum SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"
      HAVE=""; for c in $WANT; do grep -qx "$c" $O/${TAG}_counters_available.txt && HAVE="$HAVE $c"; done
      echo "counters used:$HAVE"
      SPECS=""
      for W in yelp hbm; do for P in atomic rmw; do
        EP=10; [ $W = hbm ] && EP=4
        rm -rf $O/prof_bpr_${W}_$P
        rocprofv3 --kernel-trace --stats -d $O/prof_bpr_${W}_$P -o k -- python $R/tools/prof_bpr_kernel.py $W $P $EP > $O/${TAG}_bpr_${W}_${P}_timing.log 2>&1; echo "stats $W $P exit $?"
        db=$(ls $O/prof_bpr_${W}_$P/*_results.db $O/prof_bpr_${W}_$P/*/*_results.db 2>/dev/null | head -1)
        python $R/tools/summarize_stats.py $db $O/${TAG}_bpr_${W}_${P}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/prof_bpr_kernel.py $W $P $EP   [$TAG; profiled run]" | head -4
        for c in $HAVE; do
          rm -rf $O/pmc_bpr_${W}_${P}_$c
          timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_bpr_${W}_${P}_$c -o k -- python $R/tools/prof_bpr_kernel.py $W $P $EP > $O/pmc_bpr_${W}_${P}_$c.log 2>&1 || echo "$c $W $P failed"
        done
        python $R/tools/summarize_pmc.py $O/${TAG}_bpr_counters_${W}_${P}.json $(for c in $HAVE; do echo -n "$c=$O/pmc_bpr_${W}_${P}_$c "; done) | grep -i hogwild | cut -c1-200
        rm -rf $O/pmc_bpr_${W}_${P}_* $O/prof_bpr_${W}_$P          # the rocpd databases are tens of MB each: gpurun copies back 64 MiB at most
      done; done
      rm -f $O/${TAG}_counters_list_full.txt;;
    py)
      S=${ARG%%:*}; A=""; [[ "$ARG" == *:* ]] && A=${ARG#*:}
      cd $R; timeout 3000 python tools/$S.py ${A//:/ } > $O/${TAG}_$S.log 2>&1; echo "$S exit $?"; tail -25 $O/${TAG}_$S.log | cut -c1-600; cd /tmp;;
    profpy)
      S=${ARG%%:*}; A=""; [[ "$ARG" == *:* ]] && A=${ARG#*:}
      rm -rf $O/prof_$S
      rocprofv3 --kernel-trace --stats -d $O/prof_$S -o $S -- python $R/tools/$S.py ${A//:/ } > $O/prof_$S.log 2>&1; echo "$S exit $?"
      db=$(ls $O/prof_$S/*_results.db $O/prof_$S/*/*_results.db 2>/dev/null | head -1)
      python $R/tools/summarize_stats.py $db $O/${TAG}_${S}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python tools/$S.py ${A//:/ }   [$TAG; profiled run]" | head -24;;
    *) echo "unknown step $STEP";;
  esac
  echo "[$STEP: $(( $(date +%s) - T0 )) s]"
done
