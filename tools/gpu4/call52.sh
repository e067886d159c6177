#!/bin/bash
# round 4, call 52: the whole -m gpu suite as the driver runs it, the multi-rank module last (its parent process now holds a GPU
# context from the parity modules: does that slow the ranks down on RCCL as it did on gloo?)
# (first attempt: a grep between pytest and the log file held the output back for 420 s and the box's silence guard ended the run)
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=14 -rs 2>&1 | tee gpurun_out/r04/c52_gpu_suite.log | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tail -32 | cut -c1-300
exit 0
