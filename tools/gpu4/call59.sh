#!/bin/bash
# round 4, call 59: seed 12045 (gin, eps.grad) against the float64 oracle
mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/fuzz_soak.py --ref64 12045 2>&1 | grep -v "amdgpu.ids" | tee -a gpurun_out/r04/c58_fuzz_ref64.txt | cut -c1-420
exit 0
