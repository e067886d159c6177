#!/bin/bash
# round 3, call 25 (records, final code with the single-thread eval interleave): one box — single-GPU line, emulated ranks
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python bench.py --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $O/fin2_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/fin2_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['roofline']['frac'])"
for PM in "8 gcn" "8 gcn --no-interleave --pieces-in 1" "8 graphsage" "8 graphsage2" "8 gcn --no-fused" "8 gcn --cache-input-aggregate" "8 gcn --src-split" "4 gcn" "2 gcn" "8 appnpstack"; do
  set -- $PM; P=$1; M=$2; shift 2; X="$*"; T=$(echo "P${P}_${M}_$X" | tr -d ' ' | tr -- '-' '_' | sed 's/___*/_/g; s/_$//')
  timeout -k 10 300 python bench.py --emulate-rank $P --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/fin2_emu_$T.json 2> $O/fin2_emu_$T.err
  echo "emu $PM: $(python -c "import json; d=json.loads([l for l in open('$O/fin2_emu_$T.json') if l.startswith('{')][-1]); e=d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction']; print(d['scheme'], d.get('fused_schedule'), round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'serial/exposed', round(e['serial'],2), round(e['exposed'],2))" 2>&1 | tail -1)"
done
