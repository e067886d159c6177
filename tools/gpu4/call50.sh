#!/bin/bash
# round 4, call 50: partitioned cases incl. the lopsided partition (ranks without edges / without anything to exchange) on RCCL
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -q -k "test_partitioned_hip_run_matches_single_gpu and not at_S" --durations=4 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" > gpurun_out/r04/c50_lopsided.log
tail -40 gpurun_out/r04/c50_lopsided.log | cut -c1-1200
exit 0
