#!/bin/bash
# round 4, call 57: soak of the differential fuzz beyond the suite's seeds
mkdir -p gpurun_out/r04
timeout -k 10 900 python tools/fuzz_soak.py 320 6320 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/c57_fuzz_soak.txt | tail -25 | cut -c1-420
exit 0
