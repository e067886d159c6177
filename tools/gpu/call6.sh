#!/bin/bash
# round 2, GPU call 6: bench.py --gpus N end to end with the real kernels (N ranks share the one GPU, gloo staging),
# then the FETCH / WRITE PMC passes of the final kernels for every model
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
for cfg in "2 auto" "4 2x2" "3 halo"; do
  set -- $cfg
  RGBX_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus $1 --workload S --exchange $2 --steps 3 --warmup 1 > $O/rehearse_N$1_$2.json 2> $O/rehearse_N$1_$2.err
  echo "rehearse N=$1 $2 rc=$? $(python -c "import json; d=json.load(open('$O/rehearse_N$1_$2.json')); print(d['n_gpus'], d['ranks_seen'], d['scheme'], round(d['ms_per_step'],2), d['final_losses'])" 2>&1 | tail -1)"
done
timeout -k 10 200 python bench.py --workload S --no-cpu-baseline --primary-only --steps 3 --warmup 1 > $O/rehearse_N1.json 2> $O/rehearse_N1.err
echo "single S: $(python -c "import json; d=json.load(open('$O/rehearse_N1.json')); print(round(d['ms_per_step'],2), d['final_losses'])" 2>&1 | tail -1)"
B="$GRAFT_REPO_ROOT/bench.py"
for M in gcn graphsage graphsage2 gat appnpstack; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmcf_L_${M}_$CNT -o p --output-format csv -- python3 $B --model $M --primary-only --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmcf_L_${M}_$CNT.log)
    echo "pmc $M $CNT rc=$?"
  done
done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/proff_L_gcn -o p --output-format csv -- python3 $B --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/proff_L_gcn.json 2> $GRAFT_REPO_ROOT/$O/proff_L_gcn.log)
echo "stats gcn rc=$?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/proff_L_gat -o p --output-format csv -- python3 $B --model gat --primary-only --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/proff_L_gat.json 2> $GRAFT_REPO_ROOT/$O/proff_L_gat.log)
echo "stats gat rc=$?"
