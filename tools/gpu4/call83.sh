#!/bin/bash
# round 4, call 83: the default line once more on the final code (last GPU minutes of the round)
mkdir -p gpurun_out/r04
timeout -k 10 140 python bench.py > gpurun_out/r04/c83_bench_default.json 2> gpurun_out/r04/c83_bench_default.err
echo "rc=$?"
python -c "
import json
d = json.loads(open('gpurun_out/r04/c83_bench_default.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'n_gpus')}, d['roofline']['frac'], d['roofline']['traffic'], d.get('yardstick', {}).get('avg_ms'), d.get('identical_results_same_run', {}).get('ms_per_step'))"
exit 0
