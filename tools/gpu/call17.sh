#!/bin/bash
# final-code bench lines (with sampled-logit parity) and kernel stats of the SURVEY 8f models DAGNN and GIN
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
for M in dagnn gin; do
  timeout -k 10 400 python bench.py --model $M --primary-only --steps 8 --warmup 3 > $O/bench_L_${M}_4.json 2> $O/bench_L_${M}_4.err
  echo "$M rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/bench_L_${M}_4.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['value']/1e9,2), round(d['roofline']['frac'],3), d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])" 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
for M in dagnn gin; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_${M}_4 -- python3 $GRAFT_REPO_ROOT/bench.py --model $M --primary-only --no-cpu-baseline --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_${M}_4.log 2>&1 || exit 1
done
