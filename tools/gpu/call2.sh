#!/bin/bash
# round 2, GPU call 2: GAT per-node backward tests, narrow-width timings + PMC passes, emulated-rank tables
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q -k "abi or gat or model_logits or hip_graph or next_row or partitioned" > $O/tests2.log 2>&1
echo "pytest rc=$?" | tee -a $O/tests2.log
tail -3 $O/tests2.log
timeout -k 10 200 python tools/narrow_width.py L 5 > $O/narrow_L.txt 2>&1 || exit 1
cat $O/narrow_L.txt
pmc() {  # name, counters...
  local name=$1; shift
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/pmc_narrow_$name -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/narrow_width.py L 2 > $GRAFT_REPO_ROOT/$O/pmc_narrow_$name.log 2>&1)
  echo "pmc $name rc=$?"
}
pmc ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum
pmc stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
emu() {  # P exchange [extra]
  local P=$1 X=$2; shift 2
  timeout -k 10 240 python bench.py --emulate-rank $P --exchange $X --no-cpu-baseline --primary-only --steps 5 --warmup 2 "$@" > $O/emu_P${P}_${X}.json 2> $O/emu_P${P}_${X}.err
  echo "emu P=$P $X rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu_P${P}_${X}.json')); print(d['ms_per_step'], d['emulated']['exchange_ms_per_epoch'])" 2>&1 | tail -1)"
}
emu 8 2x4
emu 8 reshard
emu 8 4x2
emu 4 reshard
emu 4 2x2
emu 2 reshard
emu 2 halo
timeout -k 10 240 python bench.py --model gat --no-cpu-baseline --primary-only --steps 5 --warmup 2 > $O/bench_L_gat_1.json 2> $O/bench_L_gat_1.err
echo "gat bench rc=$?"; python -c "import json; d=json.load(open('$O/bench_L_gat_1.json')); print(d['ms_per_step'], d['kernel_ms_by_kind'])"
