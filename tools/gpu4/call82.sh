#!/bin/bash
# round 4, call 82: tests/test_gpu_dist.py on the code as it stands (task-split grid check, nan for empty eval masks)
mkdir -p gpurun_out/r04
timeout -k 10 330 python -m pytest tests/test_gpu_dist.py -q -x --durations=4 -rs 2>&1 | tee gpurun_out/r04/c82_gpu_dist.log | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tail -10 | cut -c1-300
exit 0
