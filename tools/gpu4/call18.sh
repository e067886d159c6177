#!/bin/bash
# round 4, records on one box: emulated rank 0 of 8 (gcn, graphsage; compute-only, replayed exposure, contended with plain and
# cache-bypassing traffic), APPNP task-split roles, and the single-GPU line of the same box; then the other models' lines
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1])
    e=d.get('emulated') or {}
    c=e.get('contended') or {}
    n=e.get('contended_cache_bypassing_traffic') or {}
    print(round(d['ms_per_step'],2), 'median', round(d['median_ms_per_step'],2), 'contended', c.get('median_ms_per_step'), 'bypass', n.get('median_ms_per_step'), c.get('error'), 'yard', (d.get('yardstick') or {}).get('avg_ms'))
except Exception as ex:
    print('no line', ex)
PY
)"; }
run rec_single_L_gcn --steps 12 --warmup 3 --no-cpu-baseline --primary-only
run rec_emu8_gcn --emulate-rank 8 --steps 12 --warmup 3 --no-cpu-baseline
run rec_emu8_graphsage --emulate-rank 8 --model graphsage --steps 12 --warmup 3 --no-cpu-baseline
run rec_emu8_appnp_train --emulate-rank 8 --model appnpstack --steps 6 --warmup 2 --no-cpu-baseline
run rec_emu8_appnp_eval --emulate-rank 8 --model appnpstack --emulate-role eval --steps 6 --warmup 2 --no-cpu-baseline
run rec_emu4_gcn --emulate-rank 4 --steps 12 --warmup 3 --no-cpu-baseline
for M in graphsage graphsage2 gat appnpstack sgc gin dagnn; do
  run rec_bench_L_$M --model $M --steps 10 --warmup 3
done
