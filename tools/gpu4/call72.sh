#!/bin/bash
# round 4, call 72: the whole -m gpu suite on the code as it stands
mkdir -p gpurun_out/r04
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=10 -rs 2>&1 | tee gpurun_out/r04/c72_gpu_suite.log | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tail -26 | cut -c1-300
exit 0
