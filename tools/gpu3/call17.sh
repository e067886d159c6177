#!/bin/bash
# round 3, call 17: the whole -m gpu suite, smoke() and the default bench line on the final code
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1700 python -m pytest tests -x -q -m gpu > $O/final_tests.log 2>&1
echo "all rc=$? $(tail -1 $O/final_tests.log)"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
echo "smoke rc=$? $(tail -1 $O/final_smoke.log)"
/usr/bin/time -v timeout -k 10 600 python bench.py > $O/final_bench.json 2> $O/final_bench.err
echo "bench rc=$? wall $(grep 'Elapsed (wall' $O/final_bench.err | sed 's/.*: //')"
python -c "import json; d=json.loads([l for l in open('$O/final_bench.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), d['median_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
