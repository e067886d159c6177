#!/bin/bash
# round 4, call 19: contended emulation with its sync-only control (gcn, P = 8)
mkdir -p gpurun_out/r04
python bench.py --emulate-rank 8 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r04/rec2_emu8_gcn.json 2> gpurun_out/r04/rec2_emu8_gcn.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/rec2_emu8_gcn.json') if l.startswith('{')][-1])
e=d['emulated']
print('compute median', d['median_ms_per_step'], 'host enqueue', d['per_rank'][0]['host_enqueue_ms_per_step'])
for k in ('contended','contended_cache_bypassing_traffic','contended_control_sync_only'):
    c=e[k]; print(k, c.get('median_ms_per_step'), 'wait', c.get('exposed_exchange_wait_ms_per_step'), c.get('error'))
PY
