#!/bin/bash
# round 4, call 47: tests/test_gpu_dist.py with the S-size bitwise ahead / not-ahead cases and experiment() as two ranks for all
# five models (RCCL)
mkdir -p gpurun_out/r04
timeout -k 10 800 python -m pytest tests/test_gpu_dist.py -q --durations=8 -rs 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$" > gpurun_out/r04/c47_gpu_dist.log
tail -40 gpurun_out/r04/c47_gpu_dist.log | cut -c1-600
exit 0
