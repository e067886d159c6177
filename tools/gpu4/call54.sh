#!/bin/bash
# round 4, call 54: which GPU kernels carry RCCL's collectives when two ranks share the GPU (torch.profiler inside rank 0), and
# the cached-operand stream test with record_stream
mkdir -p gpurun_out/r04
RGBX_PROBE_PROFILE=1 timeout -k 10 240 python tools/rccl_shared_gpu_probe.py 2 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tee gpurun_out/r04/c54_rccl_kernels.txt | tail -30 | cut -c1-260
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k "cached_operands" 2>&1 | tail -2
exit 0
