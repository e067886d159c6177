#!/bin/bash
# round 4, call 36: tests/test_gpu_dist.py with the one-GPU reference runs in child processes (the pytest process holds no GPU
# context beside the ranks), then the whole -m gpu suite
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q --durations=8 2>&1 | tee gpurun_out/r04/c36_gpu_dist.log | tail -14 &&
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=12 --deselect tests/test_gpu_dist.py 2>&1 | tee gpurun_out/r04/c36_gpu_rest.log | tail -18
exit 0
