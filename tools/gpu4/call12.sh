#!/bin/bash
# round 4, call 12: gemm_tn variants: A = 768 K-slabs (3 workgroups per CU, what the registers allow), B = 4 waves per SIMD (7 spills)
mkdir -p gpurun_out/r04
for v in A B; do echo "== variant $v (B column) vs in-tree (A column)"; python tools/dense_bench.py tools/ab/gemm$v/librgbx_hip.so 2>&1 | grep "gemm_tn"; done | tee gpurun_out/r04/c12_gemm_variants.txt
exit 0
