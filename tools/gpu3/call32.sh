#!/bin/bash
# round 3, records of the final code (single-head GAT on the fused kernel; next step computed ahead): bench lines of every
# model, --stats and FETCH_SIZE / WRITE_SIZE passes of the models whose dominant kernel's source changed, emulated ranks
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout -k 10 600 python bench.py > $O/rec2_bench_L_gcn.json 2> $O/rec2_bench_L_gcn.err
echo "bench L gcn rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rec2_bench_L_gcn.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['traffic'])")"
for WM in "L gcn" "L graphsage" "L graphsage2" "L gin" "L gat" "S gcn" "S gat"; do
  set -- $WM; W=$1; M=$2
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc2_${W}_${M}_$CNT
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc2_${W}_${M}_$CNT -o p --output-format csv -- python3 $B --workload $W --model $M --primary-only --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc2_${W}_${M}_$CNT.log)
    echo "pmc $W $M $CNT rc=$?"
  done
  rm -rf $O/stats2_${W}_$M
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats2_${W}_$M -o p --output-format csv -- python3 $B --workload $W --model $M --primary-only --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/stats2_${W}_$M.json 2> $GRAFT_REPO_ROOT/$O/stats2_${W}_$M.log)
  echo "stats $W $M rc=$?"
  rm -f $O/pmc2_${W}_${M}_*/p_kernel_trace.csv $O/stats2_${W}_$M/p_kernel_trace.csv
done
