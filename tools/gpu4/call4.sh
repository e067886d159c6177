#!/bin/bash
# round 4, call 4: fused-kernel forms A/B (in-tree = round 3/4 sources vs round 2's final sources) with the plain-SpMM yardstick
# of the same box, then the whole -m gpu suite
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
python tools/ab_fused_forms.py tools/ab/r02/librgbx_hip.so L 5 2>&1 | tee $O/c4_ab_forms_r02.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/c4_gpu_suite.log | tail -15
