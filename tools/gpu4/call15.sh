#!/bin/bash
# round 4, call 15: the S-size partitioned cases in one pytest process (the order in which the APPNP reshard(4) case stalled)
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -s -k "at_S" 2>&1 | tee gpurun_out/r04/c15_distS.log | grep -a "^partitioned\|passed\|failed\|Timeout"
