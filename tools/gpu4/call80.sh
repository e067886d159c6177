#!/bin/bash
# round 4, call 80: experiment() replayed as a hipGraph vs eager over random models (calls 76 and 79 ended in the endless mask loop of infeasible splits)
mkdir -p gpurun_out/r04
timeout -k 10 700 python tools/graph_vs_eager_sweep.py 0 90 2>&1 | tee gpurun_out/r04/c80_graph_vs_eager.txt | grep -v "amdgpu.ids" | tail -16 | cut -c1-700
exit 0
