#!/bin/bash
# adaptive piece count: dist GPU tests, emulated rank 0 of 8 (must still pick 4 pieces at L without a measured latency),
# and with RGBX_LINK_LATENCY_US=300 (fewer pieces)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
emu() {  # tag [env...]
  local T=$1; shift
  env "$@" timeout -k 10 240 python bench.py --emulate-rank 8 --no-cpu-baseline --primary-only --steps 8 --warmup 3 > $O/emu27_${T}.json 2> $O/emu27_${T}.err
  echo "emu $T rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu27_${T}.json')); print(round(d['ms_per_step'],3), d['scheme'], sorted(d['emulated']['exchanges_per_epoch']))" 2>&1 | tail -1)"
}
emu default RGBX_X=1
emu lat300 RGBX_LINK_LATENCY_US=300
