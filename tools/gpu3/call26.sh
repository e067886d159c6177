#!/bin/bash
# new single-head GAT path: kernel tests, then the GAT tests that were green before
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gat or second_aggregate or edge_softmax" > gpurun_out/r03/gat1_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r03/gat1_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --model gat --primary-only --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03/gat1_bench.json 2> gpurun_out/r03/gat1_bench.err
rc=$?
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03/gat1_bench.json').read().strip().splitlines()[-1])
    print(d['ms_per_step'], d.get('median_ms_per_step'), json.dumps(d['kernel_ms_by_kind']), d.get('sampled_logit_parity'), d.get('final_losses'))
except Exception as e:
    print('bench parse failed', e)
PY
tail -5 gpurun_out/r03/gat1_bench.err
exit $rc
