#!/bin/bash
# round 4, records on one box (re-run after the last change to spmm_linear.hip: the balanced grid of dense_stream_kernel — host code, but the PMC stamp hashes the file): the default bench line, rocprofv3 --stats of the same command in its primary-only form, FETCH_SIZE /
# WRITE_SIZE passes of the models whose dominant kernel is spmm_linear_kernel (source changed this round: non-temporal stores, two
# statistics sets), SQ LDS counters of the dense launches
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
for WM in "L gcn" "L graphsage" "L graphsage2" "L gin" "S gcn"; do
  set -- $WM; W=$1; M=$2
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${W}_${M}_$CNT
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_${W}_${M}_$CNT -o p --output-format csv -- python3 $B --workload $W --model $M --primary-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_${W}_${M}_$CNT.log)
    echo "pmc $W $M $CNT rc=$?"
  done
  rm -rf $O/stats_${W}_$M
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats_${W}_$M -o p --output-format csv -- python3 $B --workload $W --model $M --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/stats_${W}_$M.json 2> $GRAFT_REPO_ROOT/$O/stats_${W}_$M.log)
  echo "stats $W $M rc=$?"
  rm -f $O/pmc_${W}_${M}_*/p_kernel_trace.csv $O/stats_${W}_$M/p_kernel_trace.csv
done
du -sh $O | tail -1
