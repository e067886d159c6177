#!/bin/bash
# DAGNN with the hop mix in rgbx_dagnn_gate_*: bench line at L (+ kernel stats), sampled-logit parity at S and L
set -e
mkdir -p gpurun_out/r02
timeout -k 10 400 python bench.py --model dagnn --primary-only --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r02/bench_L_dagnn_gate.json 2> gpurun_out/r02/bench_L_dagnn_gate.err
tail -c 1500 gpurun_out/r02/bench_L_dagnn_gate.json
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k dagnn 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof_dagnn -- python3 $GRAFT_REPO_ROOT/bench.py --model dagnn --primary-only --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r02/prof_dagnn.log 2>&1
