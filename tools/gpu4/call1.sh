#!/bin/bash
# round 4, call 1: the new parity tests (gradients at S / L, partitioned path at S, device ingest, fused Adam outside experiment())
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
python -m pytest tests/test_gpu_ingest.py -x -q -s > $O/c1_ingest.log 2>&1; echo "ingest rc=$?" | tee -a $O/c1_summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_adam_outside or transposed_weight_cache" > $O/c1_adam.log 2>&1; echo "adam rc=$?" | tee -a $O/c1_summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "gradients_at_benchmark_size" > $O/c1_grads.log 2>&1; echo "grads rc=$?" | tee -a $O/c1_summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -s -k "at_S" > $O/c1_distS.log 2>&1; echo "distS rc=$?" | tee -a $O/c1_summary.txt
tail -5 $O/c1_ingest.log $O/c1_adam.log $O/c1_grads.log $O/c1_distS.log
