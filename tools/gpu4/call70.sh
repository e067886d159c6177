#!/bin/bash
# round 4, call 71: nearly identical rows; one GATConv (8 heads x 16) on a graph with a 3,600-in-edge target and an 1,800-out-edge source: backward with
# the rows chunked (default) and unchunked, against the oracle in float64
timeout -k 10 200 python scratch/debug_gat_chunks.py 0.01 2>&1 | grep -v "amdgpu.ids" | cut -c1-200
exit 0
