#!/bin/bash
# round 4, call 14: tests/test_gpu_dist.py as a whole (the S-size APPNP reshard(4) case stalled inside the suite twice) after the
# pinned host staging of the gloo rehearsal path
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tee gpurun_out/r04/c14_gpu_dist.log | tail -5
