#!/bin/bash
# round 4, call 66: records of the GAT configs after the change to gat.hip (row-split scratch layout): FETCH_SIZE / WRITE_SIZE
# passes, rocprofv3 --stats, the bench lines (S and L)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
for WM in "L gat" "S gat"; do
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
exit 0
