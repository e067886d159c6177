#!/bin/bash
# round 4, call 61: whole-model fuzz, first soak
mkdir -p gpurun_out/r04
timeout -k 10 900 python tools/fuzz_soak.py --models 0 1500 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/c61_model_fuzz.txt | tail -30 | cut -c1-500
exit 0
