#!/bin/bash
# round 4, call 49: whole models on degenerate graphs (30 cases)
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -k "degenerate" 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/c49_degenerate.log
tail -60 gpurun_out/r04/c49_degenerate.log | cut -c1-260
exit 0
