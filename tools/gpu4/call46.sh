#!/bin/bash
# round 4, call 46: five ranks over RCCL on the one GPU (six did not fit the process guard)
mkdir -p gpurun_out/r04
run() {
  name=$1; shift
  RGBX_LINK_GBS=60 RGBX_LINK_LATENCY_US=30 timeout -k 10 400 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r04/c46_$name.json 2> gpurun_out/r04/c46_$name.err
  rc=$?
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c46_$name.json").read().strip().splitlines()[-1])
    print("$name", $rc, d.get("final_losses"), d.get("scheme"), d.get("fused_schedule"), d.get("launcher", {}).get("attempt"), d.get("error"))
except Exception as e:
    print("$name no line:", e)
PY
  return $rc
}
run S_gcn_5 --gpus 5 --workload S --model gcn &&
run L_gcn_5 --gpus 5 --workload L --model gcn &&
run L_gcn_5_reshard --gpus 5 --workload L --model gcn --exchange reshard &&
run L_appnp_5 --gpus 5 --workload L --model appnpstack
exit 0
