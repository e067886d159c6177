#!/bin/bash
# replicate scheme (dist.ReplicaGraph): GPU dist tests, emulated rank 0 of 2 / 4 / 8 against the exchange schemes
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k replicate 2>&1 | tail -3
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
emu() {  # tag P exchange [extra]
  local T=$1 P=$2 X=$3; shift 3
  timeout -k 10 240 python bench.py --emulate-rank $P --exchange $X --no-cpu-baseline --primary-only --steps 8 --warmup 3 "$@" > $O/emu20_${T}.json 2> $O/emu20_${T}.err
  echo "emu $T rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu20_${T}.json')); print(round(d['ms_per_step'],3), d['scheme'], d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction'], d.get('modelled_seconds_per_epoch_first_two_layers'))" 2>&1 | tail -1)"
}
emu P2_auto 2 auto
emu P2_reshard 2 reshard
emu P4_auto 4 auto
emu P4_replicate 4 replicate
emu P8_replicate 8 replicate
emu P2_sage 2 auto --model graphsage
