#!/bin/bash
# round 4, call 37: can RCCL run two ranks on the one GPU when each rank names a different NCCL_HOSTID?
mkdir -p gpurun_out/r04
NCCL_DEBUG=INFO timeout -k 10 200 python tools/rccl_shared_gpu_probe.py 2 > gpurun_out/r04/c37_rccl_probe.log 2>&1
echo "rc=$?"
grep -v "NCCL INFO" gpurun_out/r04/c37_rccl_probe.log | tail -30
grep -c "NCCL INFO" gpurun_out/r04/c37_rccl_probe.log
grep -i "duplicate\|NET/\|via NET\|hostHash\|Using network" gpurun_out/r04/c37_rccl_probe.log | head -12
exit 0
