#!/bin/bash
# BASELINE configs 2 and 3 (workload S: gcn, gat): kernel stats + FETCH / WRITE passes on the final code
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
for M in gcn gat; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmcs_S_${M}_$CNT
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmcs_S_${M}_$CNT -o p --output-format csv -- python3 $B --workload S --model $M --primary-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmcs_S_${M}_$CNT.log)
    rc=$?; echo "pmc $M $CNT rc=$rc"; [ $rc -eq 0 ] || exit 1
  done
  rm -rf $O/profs_S_$M
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/profs_S_$M -o p --output-format csv -- python3 $B --workload S --model $M --primary-only --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/profs_S_$M.json 2> $GRAFT_REPO_ROOT/$O/profs_S_$M.log)
  echo "stats $M rc=$?"
done
