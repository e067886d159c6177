#!/bin/bash
# round 3, call 21: SQ counters of the DENSE launches (MFMA busy cycles, LDS activity) on rank 0's share of L: is the
# return stage matrix-pipe-bound, as DESIGN.md 3.2b says?
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
pass() {
  local T=$1; shift
  rm -rf $O/sq_$T
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/sq_$T -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/fused_layer_bench.py L > /dev/null 2> $GRAFT_REPO_ROOT/$O/sq_$T.log)
  echo "pass $T rc=$?"
  rm -f $O/sq_$T/p_kernel_trace.csv
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python tools/profile_summary.py pmc $O/sq_mfma $O/sq_lds --out $O/sq_dense.csv --cmd "rocprofv3 --kernel-trace --pmc <SQ counters, two passes> -- python3 tools/fused_layer_bench.py L (rank 0 of 8's share of workload L)" --top 80 | tail -1
grep -E "dense_stream|true, true, true|gemm_tn_partial|Cijk" $O/sq_dense.csv | cut -c1-60,150-260 | head -40
