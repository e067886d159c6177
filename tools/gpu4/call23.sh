#!/bin/bash
# round 4, records on one box (re-run after the last kernel-source changes: park4 store order, spmm non-temporal stores): the default bench line, rocprofv3 --stats of the same command in its primary-only form, FETCH_SIZE /
# WRITE_SIZE passes of the models whose dominant kernel is spmm_linear_kernel (source changed this round: non-temporal stores, two
# statistics sets), SQ LDS counters of the dense launches
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/rec_bench_L_gcn.json 2> $O/rec_bench_L_gcn.err
echo "bench L gcn rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rec_bench_L_gcn.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['traffic'], round(d['yardstick']['avg_ms'],3))")"
for WM in "L gcn" "L graphsage" "L graphsage2" "L gin" "S gcn" "L appnpstack" "L sgc" "L dagnn"; do
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
# LDS counters of the dense launches (dense_stream_kernel after the permuted tile): one pass, SQ block only
rm -rf $O/pmc_dense_sq
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$O/pmc_dense_sq -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dense_bench.py 2000000 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_dense_sq.log)
echo "pmc dense sq rc=$?"
rm -f $O/pmc_dense_sq/p_kernel_trace.csv
du -sh $O | tail -1
