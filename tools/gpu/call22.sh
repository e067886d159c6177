#!/bin/bash
# secondary: the same sizes with power-law in- and out-degrees (hub rows -> row-split plans), per model
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
for M in gcn graphsage gat appnpstack; do
  timeout -k 10 400 python bench.py --model $M --degree powerlaw --primary-only --steps 8 --warmup 3 > $O/bench_L_${M}_powerlaw.json 2> $O/bench_L_${M}_powerlaw.err
  echo "$M rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/bench_L_${M}_powerlaw.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['value']/1e9,2), round(d['roofline']['frac'],3), d['config']['edges_aggregated_per_propagate'], d['kernel_ms_by_kind'], d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])" 2>&1 | tail -1)"
done
