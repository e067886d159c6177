#!/bin/bash
# round 4, call 44: (a) control - the two-stream unit test with the wait taken out must FAIL; (b) tests/test_gpu_dist.py incl. the
# bitwise two-stream eval cases on RCCL
mkdir -p gpurun_out/r04
timeout -k 10 200 python - <<'PY' 2>&1 | grep -v "amdgpu.ids" | tail -5
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from rgb_experiment_amd import ops
import test_gpu_parity as T
ops._MadeOn.get = lambda self: self.value  # the code before the fix
try:
    T.test_cached_operands_are_safe_across_streams(torch.device("cuda:0"))
    print("control: NOT detected (the test passes without the wait)")
except AssertionError as e:
    print("control: detected - without the wait the test fails:", repr(e)[:120])
PY
timeout -k 10 700 python -m pytest tests/test_gpu_dist.py -q --durations=8 -x -rs 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$" > gpurun_out/r04/c44_gpu_dist.log
tail -16 gpurun_out/r04/c44_gpu_dist.log | cut -c1-400
exit 0
