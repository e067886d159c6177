#!/bin/bash
# full GPU suite + smoke + default bench with the final round-2 code
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests9.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests9.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_L_3.json 2> $O/bench_L_3.err; echo "bench rc=$?"
python -c "import json; d=json.loads([l for l in open('$O/bench_L_3.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
