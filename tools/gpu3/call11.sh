#!/bin/bash
# round 3, call 11: cache_input_aggregate (single GPU test + dist), emulated rank numbers incl. the opt-in, default line
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cached_input or shared_eval or hip_graph or fold_bn" > $O/c11_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/c11_tests.log)"
for F in "--no-interleave" "--no-interleave --cache-input-aggregate" "--no-interleave --model graphsage" "--no-interleave --model graphsage --cache-input-aggregate"; do
  timeout -k 10 300 python bench.py --emulate-rank 8 --primary-only --no-cpu-baseline --steps 12 --warmup 3 $F > $O/c11.json 2>$O/c11.err
  python -c "import json; d=json.loads([l for l in open('$O/c11.json') if l.startswith('{')][-1]); print('$F', d.get('fused_schedule'), round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), round(d['per_rank'][0]['host_enqueue_ms_per_step'],2))"
done
timeout -k 10 600 python bench.py > $O/c11_bench.json 2> $O/c11_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/c11_bench.json') if l.startswith('{')][-1])
print('L', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],3))
for k in ('cached_input_aggregate_same_run','configs_1_same_run','configs_0_same_run'):
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in d[k].items() if a in ('ms_per_step','value','primary_ms_per_step','primary','eager_ms_per_epoch','hip_graph_ms_per_epoch','error')})
PY
