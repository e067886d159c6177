#!/bin/bash
# round 4, call 32 (the rehearsal of round 3 on this round's code: supervisors that wait for rank 0, the communicator self-test with works in flight, pinned staging, the whole-graph memory guard): `bench.py --gpus N` as typed on ONE GPU (ranks share it, gloo collectives staged through the host):
# the supervised launch end to end on the real kernels — a clean run of the fused schedule, a rank that stalls in the
# first attempt (fresh workers with the conservative flags produce the line), a rank that raises
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
export RGBX_DIST_BACKEND=gloo
show() { python -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); la=d['launcher']; print('$2', d['n_gpus'], d['ranks_seen'], d['scheme'], 'fused', d['fused_schedule'], round(d['ms_per_step'],2), 'attempt', la['attempt'], la['extra_flags'], 'fallback', None if not la['fallback'] else [f['reason'][:90] for f in la['fallback']['failed']], 'setup_s', d['per_rank'][0]['setup_s'], d['final_losses'])" 2>&1 | tail -1; }
timeout -k 10 400 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse4_clean4.json 2> $O/rehearse4_clean4.err; echo "rc=$?"; show $O/rehearse4_clean4.json clean4
timeout -k 10 400 python bench.py --gpus 2 --workload S --steps 3 --warmup 1 --no-cpu-baseline --exchange reshard > $O/rehearse4_clean2.json 2> $O/rehearse4_clean2.err; echo "rc=$?"; show $O/rehearse4_clean2.json clean2
RGBX_TEST_FAULT=stall:1:0:timed_region RGBX_LAUNCH_STALL_S=25 timeout -k 10 600 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse4_stall4.json 2> $O/rehearse4_stall4.err; echo "rc=$?"; show $O/rehearse4_stall4.json stall4
RGBX_TEST_FAULT=raise:2:0:first_epoch RGBX_LAUNCH_STALL_S=25 timeout -k 10 600 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse4_raise4.json 2> $O/rehearse4_raise4.err; echo "rc=$?"; show $O/rehearse4_raise4.json raise4
unset RGBX_DIST_BACKEND
timeout -k 10 300 python bench.py --workload S --steps 3 --warmup 1 --no-cpu-baseline --primary-only > $O/rehearse4_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/rehearse4_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['final_losses'])"
