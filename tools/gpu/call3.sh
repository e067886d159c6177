#!/bin/bash
# round 2, GPU call 3: GAT tests again, host-boundness of the emulated 8-rank epoch (pieces / interleave sweep + kernel stats)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_abi.py tests/test_gpu_parity.py -m gpu -x -q -k "abi or gat" > $O/tests3.log 2>&1
echo "pytest rc=$?"; tail -3 $O/tests3.log
emu() {  # tag P exchange [extra]
  local T=$1 P=$2 X=$3; shift 3
  timeout -k 10 240 python bench.py --emulate-rank $P --exchange $X --no-cpu-baseline --primary-only --steps 8 --warmup 3 "$@" > $O/emu3_${T}.json 2> $O/emu3_${T}.err
  echo "emu $T rc=$? $(python -c "import json,sys; d=json.load(open('$O/emu3_${T}.json')); print(round(d['ms_per_step'],3), round(d['per_rank'][0]['aggregation_ms_per_step'],3), {k:(v['n']//8, round(v['avg_ms'],3)) for k,v in d['kernel_ms_by_kind'].items()})" 2>&1 | tail -1)"
}
emu p4_il 8 2x4 --pieces 4
emu p4_seq 8 2x4 --pieces 4 --no-interleave
emu p2_il 8 2x4 --pieces 2
emu p2_seq 8 2x4 --pieces 2 --no-interleave
emu p1_seq 8 2x4 --pieces 1 --no-interleave
emu rs_p2_seq 8 reshard --pieces 2 --no-interleave
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_emu_2x4 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --emulate-rank 8 --exchange 2x4 --pieces 2 --no-interleave --no-cpu-baseline --primary-only --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_emu_2x4.json 2> $GRAFT_REPO_ROOT/$O/prof_emu_2x4.log)
echo "prof rc=$?"
python - <<'PY'
import csv,glob,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r02'
f=glob.glob(O+'/prof_emu_2x4/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[-1])))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total GPU ms',tot/1e6,'calls',calls)
for r in rows[:14]: print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
