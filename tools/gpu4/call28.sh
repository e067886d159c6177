#!/bin/bash
# round 4, call 28: the -m gpu suite with per-test durations (where the 9 minutes go)
mkdir -p gpurun_out/r04
timeout -k 10 1150 python -m pytest tests -m gpu -x -q --durations=45 2>&1 | tee gpurun_out/r04/c28_gpu_suite_durations.log | tail -60
