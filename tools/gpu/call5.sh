#!/bin/bash
# round 2, GPU call 5: full GPU suite after the BatchNorm fold / GAT changes, default bench, emulated-rank table
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests5.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_L_2.json 2> $O/bench_L_2.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('$O/bench_L_2.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['kernel_ms_by_kind'])"
for M in graphsage gat; do
  timeout -k 10 300 python bench.py --model $M --primary-only --steps 8 --warmup 3 > $O/bench_L_${M}_2.json 2> $O/bench_L_${M}_2.err
  echo "$M rc=$? $(python -c "import json; d=json.load(open('$O/bench_L_${M}_2.json')); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])" 2>&1 | tail -1)"
done
emu() {  # tag P exchange [extra]
  local T=$1 P=$2 X=$3; shift 3
  timeout -k 10 240 python bench.py --emulate-rank $P --exchange $X --no-cpu-baseline --primary-only --steps 8 --warmup 3 "$@" > $O/emu5_${T}.json 2> $O/emu5_${T}.err
  echo "emu $T rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu5_${T}.json')); print(round(d['ms_per_step'],3), d['scheme'], d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction'])" 2>&1 | tail -1)"
}
emu P8_auto 8 auto
emu P8_2x4_seq 8 2x4 --no-interleave
emu P8_reshard 8 reshard
emu P4_auto 4 auto
emu P2_auto 2 auto
emu P8_sage 8 auto --model graphsage
emu P8_appnp 8 auto --model appnpstack
