#!/bin/bash
# whole -m gpu suite on the single-head GAT path + smoke, then GAT bench lines (uniform, power-law, S)
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r03/gat2_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r03/gat2_tests.log
[ $rc -eq 0 ] || exit $rc
for A in "L uniform" "L powerlaw" "S uniform"; do
  set -- $A
  timeout -k 10 400 python bench.py --workload $1 --degree $2 --model gat --primary-only --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03/gat2_bench_$1_$2.json 2> gpurun_out/r03/gat2_bench_$1_$2.err || { echo "bench $A failed"; tail -5 gpurun_out/r03/gat2_bench_$1_$2.err; exit 1; }
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r03/gat2_bench_{sys.argv[1]}_{sys.argv[2]}.json').read().strip().splitlines()[-1])
print(sys.argv[1:], d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernel_ms_by_kind'].items()}, d['roofline']['frac'], d['roofline']['kernel'], d.get('sampled_logit_parity'))
PY
done
