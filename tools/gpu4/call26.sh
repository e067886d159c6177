#!/bin/bash
# round 4, call 26: experiment() through its defaults — the example script for four models (+ C&S), then defaults vs the
# reference-shaped epoch at S (hipGraph loop) and at L (eager loop)
mkdir -p gpurun_out/r04
for m in gcn gat appnpstack graphsage2; do python examples/simple_example.py $m 2>&1 | tail -1; done
python examples/simple_example.py gcn --cs 2>&1 | tail -1
python tools/exp_defaults_check.py S gcn 8 2>&1 | grep -v amdgpu
python tools/exp_defaults_check.py S graphsage 8 2>&1 | grep -v amdgpu
python tools/exp_defaults_check.py L gcn 6 2>&1 | grep -v amdgpu
exit 0
