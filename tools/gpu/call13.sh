#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
for M in graphsage graphsage2 gat appnpstack; do
  timeout -k 10 300 python bench.py --model $M --primary-only --steps 8 --warmup 3 > $O/bench_L_${M}_3.json 2> $O/bench_L_${M}_3.err
  echo "$M rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/bench_L_${M}_3.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['value']/1e9,2), round(d['roofline']['frac'],3), d['roofline']['traffic'] is not None, d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])" 2>&1 | tail -1)"
done
timeout -k 10 300 python bench.py --workload S --model gat --primary-only --steps 20 --warmup 5 > $O/bench_S_gat_3.json 2> $O/bench_S_gat_3.err
echo "S gat $(python -c "import json; d=json.loads([l for l in open('$O/bench_S_gat_3.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],3), round(d['value']/1e9,2), d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])" 2>&1 | tail -1)"
