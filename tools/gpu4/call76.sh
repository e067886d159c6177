#!/bin/bash
# round 4, call 79: experiment() replayed as a hipGraph vs eager over random models (call 76 ran into the reference-faithful endless mask loop at seed 3)
mkdir -p gpurun_out/r04
timeout -k 10 700 python tools/graph_vs_eager_sweep.py 0 130 2>&1 | tee gpurun_out/r04/c79_graph_vs_eager.txt | grep -v "amdgpu.ids" | tail -16 | cut -c1-700
exit 0
