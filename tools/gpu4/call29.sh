#!/bin/bash
# round 4, call 29: dense_stream_kernel with a balanced grid: dense bench vs round 3 + the dense / fused tests
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -x -q -k "dense or fused or cached" 2>&1 | tail -2
python tools/dense_bench.py tools/ab/r03/librgbx_hip.so 2>&1 | grep -v '^\[{' | tee gpurun_out/r04/c29_dense_bench.txt | grep "rows -> rows (+\|root\|blocked\|column sums"
exit 0
