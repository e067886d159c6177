#!/bin/bash
# dense layers of DAGNN / GIN on ops.linear (weight gradients from rgbx_gemm_tn_f32): tests, bench lines, GIN kernel stats
set -e
mkdir -p gpurun_out/r02
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "next_row or dagnn or hip_graph_epoch or experiment" 2>&1 | tail -3
timeout -k 10 400 python bench.py --model dagnn --primary-only --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r02/bench_L_dagnn_gate2.json 2> gpurun_out/r02/bench_L_dagnn_gate2.err
python -c "import json; d=json.loads(open('gpurun_out/r02/bench_L_dagnn_gate2.json').read().strip().splitlines()[-1]); print('dagnn', d['ms_per_step'], d['kernel_ms_by_kind'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof_gin -- python3 $GRAFT_REPO_ROOT/bench.py --model gin --primary-only --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r02/prof_gin.log 2>&1
tail -c 600 $GRAFT_REPO_ROOT/gpurun_out/r02/prof_gin.log | head -c 400
