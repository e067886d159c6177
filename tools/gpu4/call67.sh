#!/bin/bash
# round 4, call 67: whole-model fuzz, seeds 1500..3500
# (first attempt: a grep in front of tee held the output back and the box's silence guard ended the run after 420 s)
mkdir -p gpurun_out/r04
timeout -k 10 1000 python tools/fuzz_soak.py --models 1500 3500 2>&1 | tee gpurun_out/r04/c67_model_fuzz.txt | grep -v "^seed [0-9]* (\|amdgpu.ids" | tail -20 | cut -c1-600
exit 0
