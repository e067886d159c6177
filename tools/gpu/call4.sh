#!/bin/bash
# round 2, GPU call 4: GAT (scores folded) tests + bench, then kernel stats and FETCH / WRITE PMC passes for every model at L
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q -k "abi or gat or model_logits or hip_graph or partitioned" > $O/tests4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests4.log
[ $rc -eq 0 ] || exit 1
B="$GRAFT_REPO_ROOT/bench.py"
for M in gat gcn graphsage graphsage2 appnpstack; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_L_$M -o p --output-format csv -- python3 $B --model $M --primary-only --no-cpu-baseline --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_L_$M.json 2> $GRAFT_REPO_ROOT/$O/prof_L_$M.log)
  echo "stats $M rc=$? $(python -c "import json; d=json.load(open('$O/prof_L_$M.json')); print(round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['kernel'])" 2>&1 | tail -1)"
  for CNT in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT -d $GRAFT_REPO_ROOT/$O/pmc_L_${M}_$CNT -o p --output-format csv -- python3 $B --model $M --primary-only --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_L_${M}_$CNT.log)
    echo "pmc $M $CNT rc=$?"
  done
done
