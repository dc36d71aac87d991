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
#   spmm-counters        FETCH_SIZE / WRITE_SIZE (separate passes) per kernel of the LightGCN step -> <tag>_lightgcn_pmc.json
#   mfma-util            MfmaUtil per MFMA kernel (NGCF step, evaluation, SimGCL step) -> <tag>_mfma_util.json
#   stats:<m>            rocprofv3 --kernel-trace --stats of one config's step (m: lightgcn | simgcl | ngcf | eval) -> <tag>_<m>_kernel_stats.txt
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
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf $O/pmc_lg_$ctr
        rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_lg_$ctr -o lg -- python $R/tools/bench_lightgcn.py --steps 20 --shape yelp2018 > $O/pmc_lg_$ctr.log 2>&1; echo "$ctr exit $?"
      done
      python $R/tools/summarize_pmc.py $O/${TAG}_lightgcn_pmc.json FETCH_SIZE=$O/pmc_lg_FETCH_SIZE WRITE_SIZE=$O/pmc_lg_WRITE_SIZE | grep -i spmm;;
    mfma-util)
      for m in ngcf eval simgcl; do
        rm -rf $O/pmc_mfma_$m
        rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/pmc_mfma_$m -o m -- $(step_cmd $m) > $O/pmc_mfma_$m.log 2>&1; echo "$m exit $?"
      done
      python $R/tools/summarize_pmc.py $O/${TAG}_mfma_util.json MfmaUtil=$O/pmc_mfma_ngcf MfmaUtil=$O/pmc_mfma_eval MfmaUtil=$O/pmc_mfma_simgcl;;
    stats)
      rm -rf $O/prof_$ARG
      rocprofv3 --kernel-trace --stats -d $O/prof_$ARG -o $ARG -- $(step_cmd $ARG) > $O/prof_$ARG.log 2>&1; echo "$ARG exit $?"
      db=$(ls $O/prof_$ARG/*_results.db $O/prof_$ARG/*/*_results.db 2>/dev/null | head -1)
      python $R/tools/summarize_stats.py $db $O/${TAG}_${ARG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $(step_cmd $ARG)   [$TAG, Yelp2018 shape d=64; profiled run]" | head -12;;
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
