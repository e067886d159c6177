#!/bin/bash
# round 3, call 12: source-piece pipelined aggregation on the real kernels; emulated rank 0 of 8 for pieces_in 1 / 2 / 4
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -k "grid" > $O/c12_dist.log 2>&1
echo "dist rc=$? $(tail -1 $O/c12_dist.log)"
for F in "--pieces-in 1" "--pieces-in 2" "--pieces-in 4" "--pieces-in 2 --pieces 2" "--pieces-in 2 --model graphsage" "--pieces-in 2 --interleave"; do
  FF=$(echo "$F" | sed 's/--interleave//')
  NI="--no-interleave"; case "$F" in *--interleave*) NI="";; esac
  timeout -k 10 300 python bench.py --emulate-rank 8 --primary-only --no-cpu-baseline --steps 12 --warmup 3 $NI $FF > $O/c12.json 2>$O/c12.err
  python -c "import json; d=json.loads([l for l in open('$O/c12.json') if l.startswith('{')][-1]); e=d['emulated']['exchange_ms_per_epoch']['60 GB/s per link and direction']; print('$F', d.get('fused_schedule'), round(d['ms_per_step'],2), round(d['median_ms_per_step'],2), 'host', round(d['per_rank'][0]['host_enqueue_ms_per_step'],2), 'exchange ms serial/exposed', round(e['serial'],2), round(e['exposed'],2))"
done
