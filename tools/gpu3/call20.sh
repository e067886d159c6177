#!/bin/bash
# round 3, call 20 (records, final code): one box — single-GPU line, emulated rank 0 of 8 (gcn, graphsage; default and
# sequential settings), per-epoch kernel tables
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > $O/fin_bench_L_gcn.json 2> $O/fin_bench_L_gcn.err
python -c "import json; d=json.loads([l for l in open('$O/fin_bench_L_gcn.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['median_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
for PM in "gcn" "gcn --no-interleave --pieces-in 1" "graphsage" "graphsage --no-interleave --pieces-in 1" "graphsage2" "gcn --no-fused" "gcn --cache-input-aggregate"; do
  set -- $PM; M=$1; shift; X="$*"; T=$(echo "P8_${M}_$X" | tr -d ' ' | tr -- '-' '_' | sed 's/___*/_/g; s/_$//')
  timeout -k 10 300 python bench.py --emulate-rank 8 --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/fin_emu_$T.json 2> $O/fin_emu_$T.err
  echo "emu $PM: $(python -c "import json; d=json.loads([l for l in open('$O/fin_emu_$T.json') if l.startswith('{')][-1]); e=d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction']; print(d['scheme'], d.get('fused_schedule'), round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'exposed', round(e['exposed'],2), d.get('interleaved_evals'))" 2>&1 | tail -1)"
done
for M in gcn graphsage; do
for S in 9 18; do (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/fin_${M}_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --model $M --no-interleave --pieces-in 1 --no-cpu-baseline --primary-only --steps $S --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/fin_${M}_s$S.log); done
python tools/epoch_diff.py $O/fin_${M}_s9 9 $O/fin_${M}_s18 18 --out $O/fin_emu8_${M}_epoch.csv --top 60 | head -2
rm -f $O/fin_${M}_s*/p_kernel_trace.csv
done
