#!/bin/bash
# round 3, final records: bench lines of every model at L (+ S gat), emulated ranks with the schedule replay — one box
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python bench.py > $O/fin4_bench_L_gcn.json 2> $O/fin4_bench_L_gcn.err
echo "L gcn rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/fin4_bench_L_gcn.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])")"
for WM in "L graphsage" "L graphsage2" "L gat" "L appnpstack" "L sgc" "L gin" "L dagnn" "S gat"; do
  set -- $WM; W=$1; M=$2
  timeout -k 10 600 python bench.py --workload $W --model $M > $O/fin4_bench_${W}_$M.json 2> $O/fin4_bench_${W}_$M.err
  echo "$W $M rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/fin4_bench_${W}_$M.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['traffic'], d.get('sampled_logit_parity', {}).get('max_abs_diff') if isinstance(d.get('sampled_logit_parity'), dict) else d.get('sampled_logit_parity'))" 2>&1 | tail -1)"
done

