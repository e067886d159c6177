#!/bin/bash
# round 4, call 74: rgbx_fused_layer_f32 by option, random shapes
mkdir -p gpurun_out/r04
timeout -k 10 700 python tools/fuzz_soak.py --fused 0 3000 2>&1 | tee gpurun_out/r04/c74_fused_fuzz.txt | grep -v "^seed [0-9]* (\|amdgpu.ids" | tail -24 | cut -c1-520
exit 0
