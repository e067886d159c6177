#!/bin/bash
# round 4, call 9: CE epilogue with two statistics sets: did the 2 spilled dwords of the K = 128 loss forms cost anything?
mkdir -p gpurun_out/r04
python tools/ab_fused_forms.py tools/ab/r02/librgbx_hip.so L 4 2>&1 | tee gpurun_out/r04/c9_ab_forms.txt | grep -v "^{"
exit 0
