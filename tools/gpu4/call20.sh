#!/bin/bash
# round 4, call 20: plain SpMM / APPNP K-loop with non-temporal output stores (B) vs plain stores (A)
mkdir -p gpurun_out/r04
python tools/ab_lib.py tools/ab/spmm_nt/librgbx_hip.so L 4 spmm,appnp 2>&1 | tee gpurun_out/r04/c20_ab_spmm_nt.txt | grep -v amdgpu.ids
exit 0
