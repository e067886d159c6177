#!/bin/bash
# task split for APPNP at P = 8: emulated rank 0 of the training group and of the eval group, against the plain scheme
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
show() { python -c "
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
e=d['emulated']; r=e['schedule_replay']
print('$2', d['scheme'], 'split', (d.get('task_split') or {}).get('role_of_rank_0'), 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'serial@60', round(e['exchange_ms_per_epoch_serial']['60 GB/s per link and direction'],2), 'exposed@50/60/76.8', [round(v['exposed_ms_per_epoch'],2) for v in r['by_link_rate'].values()], {k:(v['n'],round(v['avg_ms'],2)) for k,v in d['kernel_ms_by_kind'].items()})
" 2>&1 | tail -1; }
for X in "--task-split on --emulate-role train" "--task-split on --emulate-role eval" "--task-split off"; do
  T=$(echo "$X" | tr -d ' ' | tr -- '-' '_')
  timeout -k 10 400 python bench.py --emulate-rank 8 --model appnpstack --primary-only --no-cpu-baseline --steps 9 --warmup 3 $X > $O/split_emu_$T.json 2> $O/split_emu_$T.err || { echo "emu $X FAILED"; tail -8 $O/split_emu_$T.err; continue; }
  show $O/split_emu_$T.json "P8 appnp $X:"
done
timeout -k 10 400 python bench.py --model appnpstack --primary-only --no-cpu-baseline --steps 5 --warmup 2 > $O/split_single_appnp.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/split_single_appnp.json') if l.startswith('{')][-1]); print('single appnp', round(d['ms_per_step'],2))"
