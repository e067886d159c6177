#!/bin/bash
# round 4, call 69: seed 2528 with the degrees capped at 1000 (no chunked rows) and at 60
for cap in 1000 60; do timeout -k 10 200 python scratch/debug_model_seed.py 2528 $cap 2>&1 | grep -v "amdgpu.ids" | grep "edges kept\|max in-degree\|convs.0\|convs.1.att\|bns.0" | cut -c1-200; done
exit 0
