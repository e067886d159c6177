#!/bin/bash
# round 3, call 13 (records): default bench line, kernel stats and FETCH / WRITE passes of the models whose dominant
# kernel lives in spmm_linear.hip (changed this round: the committed PMC figures of round 2 are stale for them)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout -k 10 600 python bench.py > $O/rec_bench_L_gcn.json 2> $O/rec_bench_L_gcn.err
echo "bench rc=$?"
for WM in "L gcn" "L graphsage" "L graphsage2" "L gin" "S gcn"; do
  set -- $WM; W=$1; M=$2
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${W}_${M}_$CNT
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_${W}_${M}_$CNT -o p --output-format csv -- python3 $B --workload $W --model $M --primary-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_${W}_${M}_$CNT.log)
    rc=$?; echo "pmc $W $M $CNT rc=$rc"
  done
  rm -rf $O/stats_${W}_$M
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats_${W}_$M -o p --output-format csv -- python3 $B --workload $W --model $M --primary-only --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/stats_${W}_$M.json 2> $GRAFT_REPO_ROOT/$O/stats_${W}_$M.log)
  echo "stats $W $M rc=$?"
  rm -f $O/pmc_${W}_${M}_*/p_kernel_trace.csv $O/stats_${W}_$M/p_kernel_trace.csv
done
