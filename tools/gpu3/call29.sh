#!/bin/bash
# next training step computed during the eval forwards (DistRunner.epoch(more=True)): GPU dist tests, emulated ranks with
# the schedule replay (exposed exchange from the trace), with and without it
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $O/ahead_tests.log 2>&1
rc=$?; tail -4 $O/ahead_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/ahead_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/ahead_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['roofline']['frac'])"
for PM in "8 gcn" "8 gcn --no-ahead" "8 gcn --no-interleave --pieces-in 1" "8 gcn --pieces-in 4" "8 graphsage" "8 graphsage --no-ahead" "8 graphsage2" "4 gcn" "4 gcn --no-ahead" "8 gcn --src-split"; do
  set -- $PM; P=$1; M=$2; shift 2; X="$*"; T=$(echo "P${P}_${M}_$X" | tr -d ' ' | tr -- '-' '_' | sed 's/___*/_/g; s/_$//')
  timeout -k 10 300 python bench.py --emulate-rank $P --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/ahead_emu_$T.json 2> $O/ahead_emu_$T.err || { echo "emu $PM FAILED"; tail -5 $O/ahead_emu_$T.err; continue; }
  echo "emu $PM: $(python -c "
import json
d=json.loads([l for l in open('$O/ahead_emu_$T.json') if l.startswith('{')][-1])
e=d['emulated']; r=e['schedule_replay']; k='60 GB/s per link and direction'
print(d['scheme'], d.get('fused_schedule'), 'ahead', d.get('next_step_ahead'), 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'tag-model exposed', round(e['exchange_ms_per_epoch'][k]['exposed'],2), 'replay: timed', round(r['timed_launch_ms_per_epoch'],2), 'exposed@50/60/76.8', [round(v['exposed_ms_per_epoch'],2) for v in r['by_link_rate'].values()], d['final_losses'])
" 2>&1 | tail -1)"
done
