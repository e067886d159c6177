#!/bin/bash
# GINConv: aggregation + root term + first Linear in one rgbx_spmm_linear_f32 launch: tests, bench line, kernel stats
set -e
mkdir -p gpurun_out/r02
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "next_row or gin" 2>&1 | tail -3
timeout -k 10 400 python bench.py --model gin --primary-only --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r02/bench_L_gin_fused.json 2> gpurun_out/r02/bench_L_gin_fused.err
python -c "import json; d=json.loads(open('gpurun_out/r02/bench_L_gin_fused.json').read().strip().splitlines()[-1]); print('gin', d['ms_per_step'], d['kernel_ms_by_kind'], d['roofline']['frac'], d.get('parity'))"
