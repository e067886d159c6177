#!/bin/bash
# rocprofv3 --kernel-trace --stats of THE default command (python3 bench.py, no flags), with the line it printed
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/stats_default
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats_default -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/stats_default.json 2> $GRAFT_REPO_ROOT/$O/stats_default.log)
echo "rc=$?"
rm -f $O/stats_default/p_kernel_trace.csv
python -c "
import json
d=json.loads([l for l in open('$O/stats_default.json') if l.startswith('{')][-1])
print(round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['kernel'], d['kernel_ms_by_kind'])"
head -4 $O/stats_default/p_kernel_stats.csv | cut -c1-200
