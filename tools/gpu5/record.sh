#!/bin/bash
# The round's measurement record for a list of workload:model:kernel triples, in one GPU-box call:
#   gpurun -- 'bash tools/gpu5/record.sh L:gcn:spmm_linear_kernel L:gat:gat_fwd_kernel'
# per triple: the bench line (all legs), rocprofv3 --kernel-trace --stats and the two --pmc passes of the primary-only command.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
here=tools/gpu5/run.sh
for t in "$@"; do
  IFS=: read -r wl model kernel <<< "$t"
  label="${wl}_${model}"
  echo "== $label: bench"
  bash $here bench "$label" --workload "$wl" --model "$model" > /dev/null || exit $?
  echo "== $label: stats"
  bash $here stats "$label" --workload "$wl" --model "$model" --primary-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null || exit $?
  echo "== $label: pmc"
  bash $here pmc "$label" "$model" "$wl" "$kernel" --workload "$wl" --model "$model" --primary-only --no-cpu-baseline --steps 5 --warmup 2 | tail -n 12 || exit $?
done
