#!/bin/bash
# round 4, call 75: every partitioned case of tests/test_gpu_dist.py again on the lopsided problem (ranks without edges) and on the
# hub problem (rows beyond the chunk threshold in the forward and the transposed CSRs), on RCCL: 24 x 2 further cases
mkdir -p gpurun_out/r04
RGBX_DIST_SWEEP=1 timeout -k 10 800 python -m pytest tests/test_gpu_dist.py -q -k "test_partitioned_hip_run_matches_single_gpu and not at_S" --durations=4 2>&1 | tee gpurun_out/r04/c75_dist_sweep.log | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tail -30 | cut -c1-1500
exit 0
