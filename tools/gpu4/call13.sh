#!/bin/bash
# round 4, call 13: the whole -m gpu suite + smoke() on the current code
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/c13_gpu_suite.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
