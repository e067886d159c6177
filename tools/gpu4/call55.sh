#!/bin/bash
# round 4, call 55: bench.py --gpus 2 as a -m gpu test, smoke()
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_dist.py -q -k "bench_gpus_2_as_typed or the_ranks_ran" -rs 2>&1 | tee gpurun_out/r04/c55.log | grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" | tail -12 | cut -c1-400
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
exit 0
