#!/bin/bash
# round 4, call 68: whole-model fuzz seed 2528 (4-layer GAT, hub graph) parameter by parameter
timeout -k 10 200 python scratch/debug_model_seed.py 2528 2>&1 | grep -v "amdgpu.ids" | tail -30 | cut -c1-200
exit 0
