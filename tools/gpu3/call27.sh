#!/bin/bash
# kernel table of the GAT epoch at L
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_gat -o p -- python3 bench.py --model gat --primary-only --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r03/prof_gat.json 2> gpurun_out/r03/prof_gat.err
rc=$?
find gpurun_out/r03/prof_gat -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03/gat_kernel_stats.csv
find gpurun_out/r03/prof_gat -name "*kernel_trace.csv" -delete
ls gpurun_out/r03/prof_gat | head
exit $rc
