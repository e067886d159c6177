#!/bin/bash
# round 3, call 8: folded eval operands — parity tests that touch eval forwards, then launches per epoch at S and C1
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $O/c8_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/c8_tests.log)"
for S in 10 20; do
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/c8_S_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload S --no-cpu-baseline --primary-only --steps $S --warmup 3 > $GRAFT_REPO_ROOT/$O/c8_S_s$S.json 2> $GRAFT_REPO_ROOT/$O/c8_S_s$S.log)
echo "prof $S rc=$?"
done
python tools/epoch_diff.py $O/c8_S_s10 10 $O/c8_S_s20 20 --out $O/c8_S_epoch.csv --top 60 | cut -c1-120 | head -8
timeout -k 10 300 python bench.py --workload S --no-cpu-baseline --steps 20 --warmup 5 > $O/c8_S.json 2>$O/c8_S.err
python -c "import json; d=json.loads([l for l in open('$O/c8_S.json') if l.startswith('{')][-1]); print('S eager', round(d['ms_per_step'],3), 'median', d['median_ms_per_step'], 'replay', d.get('hip_graph_replay'))"
