#!/bin/bash
# A/B on one box: unpack of the outbound pieces as one gather / index_select / a strided copy per piece
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
emu() {  # tag
  local T=$1; shift
  timeout -k 10 240 python bench.py --emulate-rank 8 --exchange 2x4 --no-interleave --no-cpu-baseline --primary-only --steps 10 --warmup 3 > $O/emu19_${T}.json 2> $O/emu19_${T}.err
  echo "emu $T rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu19_${T}.json')); print(round(d['ms_per_step'],3), d['scheme'])" 2>&1 | tail -1)"
}
RGBX_UNPACK=copies emu copies1
RGBX_UNPACK=gather emu gather1
RGBX_UNPACK=index_select emu isel1
RGBX_UNPACK=copies emu copies2
RGBX_UNPACK=gather emu gather2
