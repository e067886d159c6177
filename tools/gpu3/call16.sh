#!/bin/bash
# round 3, call 16 (records): per-epoch kernel tables of the emulated rank 0 of 8 on the final code (gcn, graphsage), idle time
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
for M in gcn graphsage; do
for S in 9 18; do (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/c16_${M}_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --model $M --no-interleave --pieces-in 1 --no-cpu-baseline --primary-only --steps $S --warmup 3 > $GRAFT_REPO_ROOT/$O/c16_${M}_s$S.json 2> $GRAFT_REPO_ROOT/$O/c16_${M}_s$S.log); done
python tools/epoch_diff.py $O/c16_${M}_s9 9 $O/c16_${M}_s18 18 --out $O/c16_emu8_${M}_epoch.csv --top 60 | head -3
python tools/trace_gaps.py $O/c16_${M}_s18/p_kernel_trace.csv $(python -c "print(open('$O/c16_emu8_${M}_epoch.csv').read().split(' in ')[1].split()[0])") 3 | cut -c1-200
python -c "import json; d=json.loads([l for l in open('$O/c16_${M}_s18.json') if l.startswith('{')][-1]); print('$M', round(d['ms_per_step'],2), d['median_ms_per_step'])"
rm -f $O/c16_${M}_s*/p_kernel_trace.csv
done
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/c16_single.json 2>/dev/null; python -c "import json; d=json.loads([l for l in open('$O/c16_single.json') if l.startswith('{')][-1]); print('single GPU same box', round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['traffic'])"
