#!/bin/bash
# FETCH / WRITE PMC passes for the SURVEY 8f models (sgc, gin, dagnn) at L: traffic of their dominant kernels
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
for M in sgc gin dagnn; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmch_L_${M}_$CNT
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmch_L_${M}_$CNT -o p --output-format csv -- python3 $B --model $M --primary-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmch_L_${M}_$CNT.log)
    rc=$?; echo "pmc $M $CNT rc=$rc"
    [ $rc -eq 0 ] || exit 1
  done
done
