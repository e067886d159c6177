#!/bin/bash
# step computed ahead, second build (commit + fold behind the first large launch): GPU dist tests, emulated rank with stalls by exchange
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $O/ahead2_tests.log 2>&1
rc=$?; tail -3 $O/ahead2_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/ahead2_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/ahead2_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['roofline']['frac'])"
for PM in "8 gcn" "8 gcn --no-ahead" "8 gcn --pieces-in 1" "8 graphsage" "4 gcn"; do
  set -- $PM; P=$1; M=$2; shift 2; X="$*"; T=$(echo "P${P}_${M}_$X" | tr -d ' ' | tr -- '-' '_' | sed 's/___*/_/g; s/_$//')
  timeout -k 10 300 python bench.py --emulate-rank $P --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/ahead2_emu_$T.json 2> $O/ahead2_emu_$T.err || { echo "emu $PM FAILED"; tail -5 $O/ahead2_emu_$T.err; continue; }
  echo "emu $PM: $(python -c "
import json
d=json.loads([l for l in open('$O/ahead2_emu_$T.json') if l.startswith('{')][-1])
e=d['emulated']; r=e['schedule_replay']; k='60 GB/s per link and direction'
print(d['scheme'], 'ahead', d.get('next_step_ahead'), 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'replay: timed', round(r['timed_launch_ms_per_epoch'],2), 'exposed@50/60/76.8', [round(v['exposed_ms_per_epoch'],2) for v in r['by_link_rate'].values()])
print('   stalls@60', r['by_link_rate'][k]['stalls_by_exchange_ms'])
" 2>&1 | tail -2)"
done
