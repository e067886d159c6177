#!/bin/bash
# round 4, call 60: seed 12045 step by step
timeout -k 10 200 python scratch/debug_gin_seed.py 12045 2>&1 | grep -v "amdgpu.ids" | tail -12
exit 0
