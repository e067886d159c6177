#!/bin/bash
# round 3, call 2: the fused per-rank schedule on the real kernels (ranks sharing the one GPU over gloo), then rank 0 of 8
# emulated: fused schedule vs the module path, same box
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q > $O/c2_dist.log 2>&1
echo "dist rc=$? $(tail -1 $O/c2_dist.log)"
for M in gcn graphsage; do
for F in "" "--no-fused"; do
  T=$(echo "$M$F" | tr -d ' -')
  timeout -k 10 300 python bench.py --emulate-rank 8 --model $M --primary-only --no-cpu-baseline --steps 10 --warmup 3 $F > $O/c2_emu8_$T.json 2> $O/c2_emu8_$T.err
  echo "emu8 $M $F rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/c2_emu8_$T.json') if l.startswith('{')][-1]); print(d['scheme'], d.get('fused_schedule'), round(d['ms_per_step'],2), d['median_ms_per_step'], d['emulated']['exchange_ms_per_epoch'])" 2>&1 | tail -1)"
done
done
timeout -k 10 300 python bench.py --emulate-rank 8 --primary-only --no-cpu-baseline --steps 10 --warmup 3 --no-interleave > $O/c2_emu8_gcn_seq.json 2> $O/c2_emu8_gcn_seq.err
echo "emu8 gcn sequential evals rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/c2_emu8_gcn_seq.json') if l.startswith('{')][-1]); print(d['scheme'], d.get('fused_schedule'), round(d['ms_per_step'],2), d['kernel_ms_by_kind'])" 2>&1 | tail -1)"
