#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q -k "cross_entropy or abi or hip_graph or model_logits" > $O/tests12b.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -2 $O/tests12b.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_L_3.json 2> $O/bench_L_3.err; echo "bench rc=$?"
python -c "import json; d=json.loads([l for l in open('$O/bench_L_3.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
bash tools/gpu/call10.sh
