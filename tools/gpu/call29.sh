#!/bin/bash
# SQ counters of the fused aggregate+transform kernel (GCN at L): LDS bank conflicts, MFMA busy cycles
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
pass() {  # tag counters...
  local T=$1; shift
  rm -rf $O/pmcq_$T
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/pmcq_$T -o p --output-format csv -- python3 $B --primary-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmcq_$T.log)
  echo "pass $T rc=$?"
}
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32
