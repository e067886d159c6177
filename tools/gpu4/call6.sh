#!/bin/bash
# round 4, call 6: final dense_stream_kernel (permuted tile + two-tile pipeline; K = 128 at 3 workgroups per CU): parity tests,
# dense forms against round 3's kernels, fused forms against round 2's (must be back at 1.000)
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
md5sum rgb_experiment_amd/csrc/librgbx_hip.so tools/ab/*/librgbx_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tee $O/c6_tests.log | tail -4 || exit 1
python tools/dense_bench.py tools/ab/r03/librgbx_hip.so 2>&1 | tee $O/c6_dense_bench.txt | grep -v "^\[{"
python tools/ab_fused_forms.py tools/ab/r02/librgbx_hip.so L 3 2>&1 | tee $O/c6_ab_forms.txt | grep -v "^{"
