#!/bin/bash
# task split at P = 2 (each rank on the whole graph): GPU tests, both roles emulated at L, the supervised launch rehearsed
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -k "split_by_task" > $O/split2_tests.log 2>&1
rc=$?; tail -3 $O/split2_tests.log
[ $rc -eq 0 ] || { grep -v "amdgpu.ids\|Gloo\|socket" $O/split2_tests.log | tail -40; exit $rc; }
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/split2_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/split2_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2))"
for X in "--emulate-role train" "--emulate-role eval" "--task-split off"; do
  T=$(echo "$X" | tr -d ' ' | tr -- '-' '_')
  timeout -k 10 400 python bench.py --emulate-rank 2 --primary-only --no-cpu-baseline --steps 9 --warmup 3 $X > $O/split2_emu_$T.json 2> $O/split2_emu_$T.err || { echo "emu $X FAILED"; tail -8 $O/split2_emu_$T.err; continue; }
  python -c "
import json
d=json.loads([l for l in open('$O/split2_emu_$T.json') if l.startswith('{')][-1])
print('P2 gcn $X:', d['scheme'], (d.get('task_split') or {}).get('role_of_rank_0'), 'ms', round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), {k:(v['n'],round(v['avg_ms'],2)) for k,v in d['kernel_ms_by_kind'].items()}, d['roofline']['frac'])
"
done
export RGBX_DIST_BACKEND=gloo
timeout -k 10 400 python bench.py --gpus 2 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/split2_rehearse2.json 2> $O/split2_rehearse2.err; echo "rehearsal rc=$?"
python -c "
import json
d=json.loads([l for l in open('$O/split2_rehearse2.json') if l.startswith('{')][-1])
print('gpus 2 (one GPU, gloo):', d['scheme'], d['task_split'], round(d['ms_per_step'],2), d['launcher']['attempt'], d['final_losses'])
"
