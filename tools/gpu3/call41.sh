#!/bin/bash
# per-epoch kernel tables of the emulated rank 0 of 8 under the DEFAULT schedule of the final code (step ahead, evals
# interleaved, 2 + 2 pieces): difference of two --stats runs of 18 and 9 steps; idle time between kernels
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
for M in gcn graphsage; do
for S in 9 18; do (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/fin6_${M}_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --model $M --no-cpu-baseline --primary-only --steps $S --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/fin6_${M}_s$S.log); done
python tools/epoch_diff.py $O/fin6_${M}_s9 9 $O/fin6_${M}_s18 18 --out $O/fin6_emu8_${M}_epoch.csv --top 60 | head -2
done
python tools/trace_gaps.py $O/fin6_gcn_s18/p_kernel_trace.csv 60 2>&1 | tail -4
rm -f $O/fin6_*_s*/p_kernel_trace.csv
