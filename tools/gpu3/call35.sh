#!/bin/bash
# round 3, call 35 (the same rehearsal on the final code: next step computed ahead): `bench.py --gpus N` as typed on ONE GPU (ranks share it, gloo collectives staged through the host):
# the supervised launch end to end on the real kernels — a clean run of the fused schedule, a rank that stalls in the
# first attempt (fresh workers with the conservative flags produce the line), a rank that raises
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export RGBX_DIST_BACKEND=gloo
show() { python -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); la=d['launcher']; print('$2', d['n_gpus'], d['ranks_seen'], d['scheme'], 'fused', d['fused_schedule'], round(d['ms_per_step'],2), 'attempt', la['attempt'], la['extra_flags'], 'fallback', None if not la['fallback'] else [f['reason'][:90] for f in la['fallback']['failed']], 'setup_s', d['per_rank'][0]['setup_s'], d['final_losses'])" 2>&1 | tail -1; }
timeout -k 10 400 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse2_clean4.json 2> $O/rehearse2_clean4.err; echo "rc=$?"; show $O/rehearse2_clean4.json clean4
timeout -k 10 400 python bench.py --gpus 2 --workload S --steps 3 --warmup 1 --no-cpu-baseline --exchange reshard > $O/rehearse2_clean2.json 2> $O/rehearse2_clean2.err; echo "rc=$?"; show $O/rehearse2_clean2.json clean2
RGBX_TEST_FAULT=stall:1:0:timed_region RGBX_LAUNCH_STALL_S=25 timeout -k 10 600 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse2_stall4.json 2> $O/rehearse2_stall4.err; echo "rc=$?"; show $O/rehearse2_stall4.json stall4
RGBX_TEST_FAULT=raise:2:0:first_epoch RGBX_LAUNCH_STALL_S=25 timeout -k 10 600 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse2_raise4.json 2> $O/rehearse2_raise4.err; echo "rc=$?"; show $O/rehearse2_raise4.json raise4
unset RGBX_DIST_BACKEND
timeout -k 10 300 python bench.py --workload S --steps 3 --warmup 1 --no-cpu-baseline --primary-only > $O/rehearse2_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/rehearse2_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['final_losses'])"
for X in "--degree powerlaw --model gcn" "--degree powerlaw --model graphsage" "--workload S --model gcn"; do
  T=$(echo "$X" | tr -d ' ' | tr -- '-' '_')
  timeout -k 10 300 python bench.py --emulate-rank 8 --primary-only --no-cpu-baseline --steps 6 --warmup 2 $X > $O/extra_emu_$T.json 2> $O/extra_emu_$T.err || { echo "emu $X FAILED"; tail -5 $O/extra_emu_$T.err; continue; }
  echo "emu $X: $(python -c "import json; d=json.loads([l for l in open('$O/extra_emu_$T.json') if l.startswith('{')][-1]); r=d['emulated']['schedule_replay']['by_link_rate']['60 GB/s per link and direction']; print(d['scheme'], d['fused_schedule'], d['next_step_ahead'], round(d['ms_per_step'],2), 'exposed@60', round(r['exposed_ms_per_epoch'],2), d['final_losses'])" 2>&1 | tail -1)"
done
