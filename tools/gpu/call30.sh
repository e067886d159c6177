#!/bin/bash
# kernel stats of the emulated ranks: rank 0 of 8 (grid 2x4) and rank 0 of 2 (replicate)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
for P in 8 2; do
  rm -rf $O/profe_P$P
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/profe_P$P -o p --output-format csv -- python3 $B --emulate-rank $P --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/profe_P$P.json 2> $GRAFT_REPO_ROOT/$O/profe_P$P.log)
  echo "stats P=$P rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/profe_P$P.json') if l.startswith('{')][-1]); print(d['scheme'], round(d['ms_per_step'],2))")"
done
