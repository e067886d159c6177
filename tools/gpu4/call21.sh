#!/bin/bash
# round 4, call 21: GAT kernels with non-temporal output stores (B) vs plain (A); dense bench after the park4 store order (LDS counters)
mkdir -p gpurun_out/r04
python tools/ab_lib.py tools/ab/gat_nt/librgbx_hip.so L 4 gat 2>&1 | tee gpurun_out/r04/c21_ab_gat_nt.txt | grep -v amdgpu.ids
export TMPDIR=/tmp
rm -rf gpurun_out/r04/pmc_dense_sq2
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/r04/pmc_dense_sq2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dense_bench.py 2000000 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r04/pmc_dense_sq2.log); echo "pmc rc=$?"
rm -f gpurun_out/r04/pmc_dense_sq2/p_kernel_trace.csv
python tools/dense_bench.py 2>&1 | grep -v '^\[{' | grep "rows -> rows (+\|root" 
exit 0
