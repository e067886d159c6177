#!/bin/bash
# round 2, GPU call 8: new tests (f-row models at full size, W^T cache), bench lines of the f-row models
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "sgc or gin or dagnn or transposed_weight or next_row or hip_graph" > $O/tests8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests8.log
[ $rc -eq 0 ] || exit 1
for M in sgc gin dagnn; do
  timeout -k 10 300 python bench.py --model $M --primary-only --steps 5 --warmup 2 > $O/bench_L_${M}.json 2> $O/bench_L_${M}.err
  echo "$M rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/bench_L_${M}.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['value']/1e9,2), round(d['roofline']['frac'],3), d['roofline']['kernel'], d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])" 2>&1 | tail -1)"
done
