#!/bin/bash
# round 4, call 35: tests/test_gpu_dist.py with the small partitioned cases merged into one spawn per world size
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q --durations=6 2>&1 | tee gpurun_out/r04/c35_gpu_dist.log | tail -12
