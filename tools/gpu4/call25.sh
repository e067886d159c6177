#!/bin/bash
# round 4, call 25: new edge-case tests
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity.py -x -q -k "largest_node_count or two_statistics_sets or two_masks or dense or fused_kernel_tiny" 2>&1 | tee gpurun_out/r04/c25_tests.log | tail -5
