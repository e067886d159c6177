#!/bin/bash
# one gather instead of a strided copy per outbound piece: dist GPU tests, emulated rank 0 of 8 (grid 2x4)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
emu() {  # tag P exchange [extra]
  local T=$1 P=$2 X=$3; shift 3
  timeout -k 10 240 python bench.py --emulate-rank $P --exchange $X --no-cpu-baseline --primary-only --steps 8 --warmup 3 "$@" > $O/emu18_${T}.json 2> $O/emu18_${T}.err
  echo "emu $T rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu18_${T}.json')); print(round(d['ms_per_step'],3), d['scheme'], d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction'])" 2>&1 | tail -1)"
}
emu P8_auto 8 auto
emu P8_2x4_seq 8 2x4 --no-interleave
emu P4_auto 4 auto
