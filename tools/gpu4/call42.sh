#!/bin/bash
# round 4, call 42: the 3-rank replicate run's TEST loss (4.854386) differs from every other scheme's (4.85271): where from?
mkdir -p gpurun_out/r04
run() {
  name=$1; shift
  RGBX_LINK_GBS=60 RGBX_LINK_LATENCY_US=30 timeout -k 10 400 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r04/c42_$name.json 2> gpurun_out/r04/c42_$name.err
  rc=$?
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c42_$name.json").read().strip().splitlines()[-1])
    print("$name", $rc, d.get("final_losses"), d.get("scheme"), d.get("error"))
except Exception as e:
    print("$name no line:", e)
PY
}
run L3_nccl --gpus 3 --workload L --model gcn
run L3_nccl_again --gpus 3 --workload L --model gcn
RGBX_DIST_BACKEND=gloo run L3_gloo --gpus 3 --workload L --model gcn
run L3_nccl_nointerleave --gpus 3 --workload L --model gcn --no-interleave
run L3_nccl_noahead --gpus 3 --workload L --model gcn --no-ahead
run L4_replicate --gpus 4 --workload L --model gcn --exchange replicate
run L2_replicate --gpus 2 --workload L --model gcn --exchange replicate --task-split off
run S3_nccl --gpus 3 --workload S --model gcn
run L3_reshard --gpus 3 --workload L --model gcn --exchange reshard
exit 0
