#!/bin/bash
# round 4, call 58: the soak's two seeds against the oracle in float64; then a further soak (6320..14320)
mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/fuzz_soak.py --ref64 731 767 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/c58_fuzz_ref64.txt | cut -c1-420
timeout -k 10 900 python tools/fuzz_soak.py 6320 14320 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/c58_fuzz_soak2.txt | tail -14 | cut -c1-420
exit 0
