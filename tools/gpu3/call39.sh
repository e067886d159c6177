#!/bin/bash
# outbound piece count under the step-ahead schedule: rank compute + replayed exposure at 1 / 2 / 4 pieces
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/pc_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/pc_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2))"
for X in "--pieces 1" "--pieces 2" "--pieces 4" "--pieces 2 --pieces-in 3" "--pieces 8"; do
  T=$(echo "$X" | tr -d ' ' | tr -- '-' '_')
  for M in gcn graphsage; do
  timeout -k 10 300 python bench.py --emulate-rank 8 --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/pc_emu_${M}_$T.json 2> $O/pc_emu_${M}_$T.err || { echo "emu $X FAILED"; tail -5 $O/pc_emu_${M}_$T.err; continue; }
  python -c "
import json
d=json.loads([l for l in open('$O/pc_emu_${M}_$T.json') if l.startswith('{')][-1])
e=d['emulated']; r=e['schedule_replay']
ex=[round(v['exposed_ms_per_epoch'],2) for v in r['by_link_rate'].values()]; ex30=[round(v['exposed_ms_per_epoch'],2) for v in e['schedule_replay_30us_per_exchange'].values()]
print('$M $X:', 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'exposed@50/60/76.8', ex, '30us', ex30, 'total@60', round(d['ms_per_step']+ex[1],2), 'total@60+30us', round(d['ms_per_step']+ex30[1],2), 'total@50', round(d['ms_per_step']+ex[0],2))
"
  done
done
