#!/bin/bash
# round 3, call 23: FETCH_SIZE / WRITE_SIZE of the emulated rank 0 of 8's kernels (fused schedule): does a rank move more
# than its algorithmic bytes?
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_emu8_$CNT
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_emu8_$CNT -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --no-interleave --pieces-in 1 --primary-only --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_emu8_$CNT.log)
  echo "pmc $CNT rc=$?"
  rm -f $O/pmc_emu8_$CNT/p_kernel_trace.csv
done
python tools/profile_summary.py pmc $O/pmc_emu8_FETCH_SIZE $O/pmc_emu8_WRITE_SIZE --out $O/emu8_pmc_hbm.csv --cmd "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --emulate-rank 8 --no-interleave --pieces-in 1 --primary-only --no-cpu-baseline --steps 4 --warmup 2 (round 3)" --top 24 | tail -1
cut -c1-70,120-260 $O/emu8_pmc_hbm.csv | head -30
