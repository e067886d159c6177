#!/bin/bash
# round 4, call 34: the whole -m gpu suite + smoke() + the default bench line on the final code (fresh PMC stamps)
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/c34_gpu_suite.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/final2_bench_L_gcn.json 2> $O/final2_bench_L_gcn.err; echo "bench rc=$?"
python -c "
import json
d=json.loads([l for l in open('$O/final2_bench_L_gcn.json') if l.startswith('{')][-1])
print(round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:40], 'yard', round(d['yardstick']['avg_ms'],3), 'ident', d.get('epochs_per_s_identical_results'))"
