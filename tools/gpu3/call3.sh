#!/bin/bash
# round 3, call 3: where the emulated rank's epoch goes (difference of two --stats runs) + fused-layer microbenchmarks
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python tools/fused_layer_bench.py L > $O/c3_flb.txt 2> $O/c3_flb.err
echo "flb rc=$?"; cat $O/c3_flb.txt
for S in 10 20; do
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/c3_prof_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --no-interleave --no-cpu-baseline --primary-only --steps $S --warmup 3 > $GRAFT_REPO_ROOT/$O/c3_prof_s$S.json 2> $GRAFT_REPO_ROOT/$O/c3_prof_s$S.log)
echo "prof $S rc=$?"
done
python tools/epoch_diff.py $O/c3_prof_s10 10 $O/c3_prof_s20 20 --out $O/c3_emu8_epoch.csv | cut -c1-150
