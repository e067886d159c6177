#!/bin/bash
# round 4, call 43: cached operands taken on another stream wait for their making (ops._MadeOn): the unit test, then the
# replicate runs whose TEST loss was off under RCCL (expected now: 4.852717 as on gloo / without the second thread)
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "cached_operands_are_safe or fused_adam_outside or two_masks_from_one" 2>&1 | tail -4
run() {
  name=$1; shift
  RGBX_LINK_GBS=60 RGBX_LINK_LATENCY_US=30 timeout -k 10 400 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r04/c43_$name.json 2> gpurun_out/r04/c43_$name.err
  rc=$?
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c43_$name.json").read().strip().splitlines()[-1])
    print("$name", $rc, d.get("final_losses"), d.get("scheme"), d.get("error"))
except Exception as e:
    print("$name no line:", e)
PY
}
run L3_nccl --gpus 3 --workload L --model gcn
run L3_nccl_again --gpus 3 --workload L --model gcn
run L2_replicate --gpus 2 --workload L --model gcn --exchange replicate --task-split off
run L4_replicate --gpus 4 --workload L --model gcn --exchange replicate
run L3_graphsage_replicate --gpus 3 --workload L --model graphsage --exchange replicate
RGBX_DIST_BACKEND=gloo run L3_graphsage_replicate_gloo --gpus 3 --workload L --model graphsage --exchange replicate
exit 0
