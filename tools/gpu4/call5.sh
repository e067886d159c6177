#!/bin/bash
# round 4, call 5: permuted LDS tile + pipelined dense_stream_kernel: parity tests of the fused / dense kernels first, then
# dense forms A (new) vs B (round 3 kernels), A vs the 3-waves-per-SIMD build, then the fused forms against round 2's kernels
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
md5sum rgb_experiment_amd/csrc/librgbx_hip.so tools/ab/*/librgbx_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or dense or cached or hip_graph or model_logits or loss_epilogue or cross_entropy or batchnorm_folded or gat_single_head or second_aggregate" 2>&1 | tee $O/c5_tests.log | tail -5 || exit 1
python tools/dense_bench.py tools/ab/r03/librgbx_hip.so 2>&1 | tee $O/c5_dense_bench.txt | grep -v "^\["
python tools/dense_bench.py tools/ab/dense_occ3/librgbx_hip.so 2>&1 | tee $O/c5_dense_bench_occ3.txt | grep "rows -> rows (\|blocked -> rows\|rows -> blocked\|column sums"
python tools/ab_fused_forms.py tools/ab/r02/librgbx_hip.so L 4 2>&1 | tee $O/c5_ab_forms.txt | grep -v "^{"
