#!/bin/bash
# round 3, final emulated records (auto piece count capped at 2 under the interleaved schedule; task split roles) — one box
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/fin5_bench_L_gcn.json 2> $O/fin5_bench_L_gcn.err
echo "L gcn rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/fin5_bench_L_gcn.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), d['roofline']['frac'])")"
for PM in "8 gcn" "8 gcn --no-ahead" "8 gcn --no-interleave --pieces-in 1" "8 graphsage" "8 graphsage2" "8 gcn --no-fused" "8 gcn --cache-input-aggregate" "8 gcn --src-split" "4 gcn" "8 appnpstack --task-split off" "8 appnpstack --emulate-role train" "8 appnpstack --emulate-role eval" "2 gcn --emulate-role train" "2 gcn --emulate-role eval" "2 gcn --task-split off" "8 gcn --pieces 4" "8 gat"; do
  set -- $PM; P=$1; M=$2; shift 2; X="$*"; T=$(echo "P${P}_${M}_$X" | tr -d ' ' | tr -- '-' '_' | sed 's/___*/_/g; s/_$//')
  timeout -k 10 300 python bench.py --emulate-rank $P --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/fin5_emu_$T.json 2> $O/fin5_emu_$T.err || { echo "emu $PM FAILED"; tail -5 $O/fin5_emu_$T.err; continue; }
  echo "emu $PM: $(python -c "
import json
d=json.loads([l for l in open('$O/fin5_emu_$T.json') if l.startswith('{')][-1])
e=d['emulated']; r=e['schedule_replay']; k='60 GB/s per link and direction'
print(d['scheme'], d.get('fused_schedule'), 'ahead', d.get('next_step_ahead'), 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'serial', round(e['exchange_ms_per_epoch_serial'][k],2), 'exposed@50/60/76.8', [round(v['exposed_ms_per_epoch'],2) for v in r['by_link_rate'].values()], '30us:', [round(v['exposed_ms_per_epoch'],2) for v in e['schedule_replay_30us_per_exchange'].values()])
" 2>&1 | tail -1)"
done
