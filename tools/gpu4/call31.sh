#!/bin/bash
# round 4, call 31: tests/test_gpu_fullsize.py after the trims and the shared oracle CSRs
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -x -q --durations=8 2>&1 | tee gpurun_out/r04/c31_fullsize.log | tail -14
