#!/bin/bash
# round 4, call 38: tests/test_gpu_dist.py with the ranks on RCCL (NCCL_HOSTID per rank, socket transport) instead of gloo
mkdir -p gpurun_out/r04
RGBX_TEST_BACKEND=rccl RGBX_TEST_DUMP_AFTER=240 timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -q --durations=8 -x 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$" > gpurun_out/r04/c38_gpu_dist_rccl.log
tail -40 gpurun_out/r04/c38_gpu_dist_rccl.log | cut -c1-400
exit 0
