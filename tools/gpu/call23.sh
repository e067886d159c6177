#!/bin/bash
# bench.py --gpus N as typed, ranks sharing the one GPU over gloo (host-staged collectives): the replicate pick at 2
# ranks and the exchange schemes at 3 / 4 ranks on the real kernels, end to end through the launcher
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
export RGBX_DIST_BACKEND=gloo
for N in 2 3 4; do
  timeout -k 10 300 python bench.py --gpus $N --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse23_$N.json 2> $O/rehearse23_$N.err
  echo "gpus=$N rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rehearse23_$N.json') if l.startswith('{')][-1]); print(d['n_gpus'], d['ranks_seen'], d['scheme'], round(d['ms_per_step'],2), d['final_losses'], d.get('modelled_seconds_per_epoch_first_two_layers'))" 2>&1 | tail -1)"
done
timeout -k 10 300 python bench.py --gpus 2 --workload S --steps 3 --warmup 1 --no-cpu-baseline --model graphsage --exchange replicate > $O/rehearse23_sage.json 2> $O/rehearse23_sage.err
echo "sage rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rehearse23_sage.json') if l.startswith('{')][-1]); print(d['scheme'], round(d['ms_per_step'],2), d['final_losses'])" 2>&1 | tail -1)"
timeout -k 10 300 python bench.py --workload S --steps 3 --warmup 1 --no-cpu-baseline --primary-only > $O/rehearse23_single.json 2>/dev/null
python -c "import json; d=json.loads([l for l in open('$O/rehearse23_single.json') if l.startswith('{')][-1]); print('single', round(d['ms_per_step'],2), d['final_losses'])"
