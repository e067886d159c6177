#!/bin/bash
# round 4, call 27: experiment() defaults vs the reference-shaped epoch, warmed up, more epochs
mkdir -p gpurun_out/r04
{ python tools/exp_defaults_check.py S gcn 200; python tools/exp_defaults_check.py L gcn 20; python tools/exp_defaults_check.py L graphsage 20; } 2>&1 | grep -v amdgpu | tee gpurun_out/r04/c27_experiment_defaults.txt
exit 0
