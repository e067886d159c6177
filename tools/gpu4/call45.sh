#!/bin/bash
# round 4, call 45: the supervised launch's failure handling on RCCL (ranks sharing the one GPU): a rank that stalls in the timed
# region of the first attempt (its peers then sit in an RCCL collective; the supervisors must end them and fresh workers with the
# conservative flags must produce the line), and a rank that raises in the first epoch
set -u
O=gpurun_out/r04; mkdir -p $O
show() { python -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); la=d['launcher']; print('$2', d['n_gpus'], d.get('ranks_seen'), d.get('scheme'), 'fused', d.get('fused_schedule'), d.get('ms_per_step') and round(d['ms_per_step'],2), 'attempt', la['attempt'], la['extra_flags'], 'fallback', None if not la['fallback'] else [f['reason'][:90] for f in la['fallback']['failed']], d.get('final_losses'), d.get('error'))" 2>&1 | tail -1; }
RGBX_LINK_GBS=60 RGBX_TEST_FAULT=stall:1:0:timed_region RGBX_LAUNCH_STALL_S=25 timeout -k 10 420 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/c45_rccl_stall4.json 2> $O/c45_rccl_stall4.err; echo "rc=$?"; show $O/c45_rccl_stall4.json stall4 &&
RGBX_LINK_GBS=60 RGBX_TEST_FAULT=raise:2:0:first_epoch RGBX_LAUNCH_STALL_S=25 timeout -k 10 420 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/c45_rccl_raise4.json 2> $O/c45_rccl_raise4.err; echo "rc=$?"; show $O/c45_rccl_raise4.json raise4
timeout -k 10 120 python -c "import torch; x=torch.ones(1<<20,device='cuda'); print('GPU alive after the faults:', float(x.sum()))" 2>&1 | tail -1
exit 0
