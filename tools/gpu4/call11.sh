#!/bin/bash
# round 4, call 11: pipelined ROOT form of dense_stream_kernel: dense bench vs round 3 (tests: 275 passed in the first attempt)
mkdir -p gpurun_out/r04
python tools/dense_bench.py tools/ab/r03/librgbx_hip.so 2>&1 | tee gpurun_out/r04/c11_dense_bench.txt | grep -v "^\[{" | grep "root\|rows -> rows (+"
exit 0
