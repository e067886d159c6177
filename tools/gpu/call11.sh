#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column_sums or folded or model_logits or fused or batchnorm" > $O/tests11.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -2 $O/tests11.log
[ $rc -eq 0 ] || exit 1
bash tools/gpu/call10.sh
