#!/bin/bash
# final check of the default N > 1 paths on one GPU after the last schedule change: GPU dist tests, emulated rank 0 of 8
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -q -m gpu > $O/last_tests.log 2>&1
rc=$?; tail -2 $O/last_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/last_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/last_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2))"
for PM in "8 gcn" "8 graphsage" "4 gcn"; do
  set -- $PM; P=$1; M=$2
  timeout -k 10 300 python bench.py --emulate-rank $P --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 > $O/last_emu_P${P}_$M.json 2> $O/last_emu_P${P}_$M.err || { echo "emu $PM FAILED"; tail -5 $O/last_emu_P${P}_$M.err; continue; }
  python -c "
import json
d=json.loads([l for l in open('$O/last_emu_P${P}_$M.json') if l.startswith('{')][-1])
e=d['emulated']; r=e['schedule_replay']
print('emu $PM:', 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'exposed@50/60/76.8', [round(v['exposed_ms_per_epoch'],2) for v in r['by_link_rate'].values()], '30us', [round(v['exposed_ms_per_epoch'],2) for v in e['schedule_replay_30us_per_exchange'].values()], d['final_losses'])
"
done
