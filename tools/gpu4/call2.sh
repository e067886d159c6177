#!/bin/bash
# round 4, call 2: fixed tests of call 1 + diagnosis of the APPNP reshard(4) case at S
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_adam_outside or transposed_weight_cache" 2>&1 | tee $O/c2_adam.log | tail -3
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -q -s -k "gradients_at_benchmark_size" 2>&1 | tee $O/c2_grads.log | grep -a "gradient parity\|passed\|failed"
RGBX_TEST_DUMP_AFTER=150 timeout -k 10 330 python -m pytest tests/test_gpu_dist.py -x -q -s -k "at_S and appnp" > $O/c2_distS.log 2>&1; echo "distS rc=$?"
tail -60 $O/c2_distS.log
