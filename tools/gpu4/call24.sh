#!/bin/bash
# round 4, FINAL records on one box (final kernel sources, fresh PMC stamps): default line, the other models' lines, the single-GPU
# line in its primary-only form + emulated rank 0 of 8 (gcn)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1])
    e=d.get('emulated') or {}
    print(round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'] or 0,3), 'traffic', d['roofline']['traffic'], 'yard', round((d.get('yardstick') or {}).get('avg_ms') or 0,3), 'ident', d.get('epochs_per_s_identical_results'), 'contended', (e.get('contended') or {}).get('median_ms_per_step'), 'ctrl', (e.get('contended_control_sync_only') or {}).get('median_ms_per_step'))
except Exception as ex:
    print('no line', ex)
PY
)"; }
run final_bench_L_gcn --steps 20 --warmup 5
run final_single_L_gcn --steps 12 --warmup 3 --no-cpu-baseline --primary-only
run final_emu8_gcn --emulate-rank 8 --steps 12 --warmup 3 --no-cpu-baseline
for M in graphsage graphsage2 gat appnpstack sgc gin dagnn; do
  run final_bench_L_$M --model $M --steps 10 --warmup 3
done
run final_bench_S_gcn --workload S --steps 30 --warmup 5
run final_bench_S_gat --workload S --model gat --steps 30 --warmup 5
