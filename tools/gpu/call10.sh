#!/bin/bash
# FETCH / WRITE PMC passes + kernel stats of the final fused kernel (gcn / graphsage / graphsage2 at L)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
for M in gcn graphsage graphsage2; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmcg_L_${M}_$CNT
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmcg_L_${M}_$CNT -o p --output-format csv -- python3 $B --model $M --primary-only --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmcg_L_${M}_$CNT.log)
    echo "pmc $M $CNT rc=$?"
  done
  rm -rf $O/profg_L_$M
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/profg_L_$M -o p --output-format csv -- python3 $B --model $M --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/profg_L_$M.json 2> $GRAFT_REPO_ROOT/$O/profg_L_$M.log)
  echo "stats $M rc=$?"
done
