#!/bin/bash
# round 4, call 16: the whole -m gpu suite + smoke() on the final code
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 1150 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/c16_gpu_suite.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
