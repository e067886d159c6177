#!/bin/bash
# round 4, call 77: the sweep's first case under a 40 s stack dump
timeout -k 10 120 python scratch/debug_sweep_case.py 0 2>&1 | grep -v "amdgpu.ids" | tail -60 | cut -c1-220
exit 0
