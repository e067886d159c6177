#!/bin/bash
# final check of the round: full GPU suite, smoke(), default bench (with its wall time), peak HBM per model
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests26.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -2 $O/tests26.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
T0=$(date +%s)
timeout -k 10 400 python bench.py > $O/bench_L_10.json 2> $O/bench_L_10.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 ))s"
python -c "import json; d=json.load(open('$O/bench_L_10.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['hbm_allocated_peak_gb'], [k for k in d if isinstance(d[k], dict) and 'error' in d[k]])"
for M in gat appnpstack dagnn; do
  timeout -k 10 300 python bench.py --model $M --primary-only --no-cpu-baseline --steps 3 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M', round(d['ms_per_step'],2), 'peak GB', round(d['hbm_allocated_peak_gb'],1))"
done
