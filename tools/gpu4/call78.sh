#!/bin/bash
# round 4, call 78: the sweep's cases 1..4, each under a 40 s stack dump
for s in 1 2 3 4; do echo "== case $s"; timeout -k 10 90 python scratch/debug_sweep_case.py $s 2>&1 | grep -v "amdgpu.ids" | tail -45 | cut -c1-200; done
exit 0
