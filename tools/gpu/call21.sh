#!/bin/bash
# full GPU suite + smoke + default bench on the code with the DAGNN / GIN kernels and the replicate scheme
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests21.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests21.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > $O/bench_L_6.json 2> $O/bench_L_6.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('$O/bench_L_6.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['parity'])"
