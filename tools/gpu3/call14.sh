#!/bin/bash
# round 3, call 14 (records): bench lines of the other models (BASELINE configs 3-5 in their one-GPU form and the SURVEY 8f
# rows), emulated rank 0 of P = 2 / 4 / 8, fused-layer microbenchmarks
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
for M in graphsage graphsage2 gat appnpstack sgc gin dagnn; do
  timeout -k 10 600 python bench.py --model $M --primary-only > $O/rec_bench_L_$M.json 2> $O/rec_bench_L_$M.err
  echo "$M rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rec_bench_L_$M.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['traffic'], d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'], (d.get('parity_k10_whole_graph') or {}).get('max_abs_diff_hip_vs_cpu'))" 2>&1 | tail -1)"
done
timeout -k 10 600 python bench.py --model gat --workload S --primary-only > $O/rec_bench_S_gat.json 2> $O/rec_bench_S_gat.err
echo "S gat rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rec_bench_S_gat.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],3), d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'])")"
for PM in "8 gcn" "8 graphsage" "8 graphsage2" "8 appnpstack" "4 gcn" "2 gcn" "8 gcn --no-fused" "8 gcn --cache-input-aggregate" "8 gcn --src-split" "8 gcn --pieces-in 1 --no-interleave" "8 graphsage --pieces-in 1 --no-interleave" "4 gcn --pieces-in 1 --no-interleave"; do
  set -- $PM; P=$1; M=$2; shift 2; X="$*"; T=$(echo "P${P}_${M}_$X" | tr -d ' ' | tr -- '-' '_')
  timeout -k 10 300 python bench.py --emulate-rank $P --model $M --primary-only --no-cpu-baseline --steps 12 --warmup 3 $X > $O/rec_emu_$T.json 2> $O/rec_emu_$T.err
  echo "emu $PM rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rec_emu_$T.json') if l.startswith('{')][-1]); e=d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction']; print(d['scheme'], d.get('fused_schedule'), round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'exchange serial/exposed', round(e['serial'],2), round(e['exposed'],2))" 2>&1 | tail -1)"
done
timeout -k 10 300 python tools/fused_layer_bench.py L > $O/rec_fused_layer_bench.txt 2>/dev/null; tail -13 $O/rec_fused_layer_bench.txt
