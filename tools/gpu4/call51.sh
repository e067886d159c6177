#!/bin/bash
# round 4, call 51: host time to issue one exchange on RCCL, by form (2 and 4 ranks sharing the GPU)
mkdir -p gpurun_out/r04
for n in 2 4; do
  timeout -k 10 280 python tools/rccl_issue_cost.py $n 300 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tee -a gpurun_out/r04/c51_rccl_issue_cost.txt | tail -8
done
exit 0
