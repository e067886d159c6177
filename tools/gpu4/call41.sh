#!/bin/bash
# round 4, call 41: further schemes on four / three ranks over RCCL on the one GPU
mkdir -p gpurun_out/r04
run() {
  name=$1; shift
  RGBX_LINK_GBS=60 RGBX_LINK_LATENCY_US=30 timeout -k 10 400 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r04/c41_$name.json 2> gpurun_out/r04/c41_$name.err
  rc=$?
  echo "== $name rc=$rc"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c41_$name.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "error", "final_losses", "fused_schedule")})
    print("  ", d.get("config", {}).get("parallelism"))
    print("  ", d.get("launcher"))
except Exception as e:
    print("no line:", e)
PY
  grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" gpurun_out/r04/c41_$name.err | tail -5 | cut -c1-300
  return $rc
}
# (six ranks: the box's process guard ended the run - 7 processes had the GPU open, limit 6; four ranks is what fits)
run L_gcn_4_2x2 --gpus 4 --workload L --model gcn --exchange 2x2 &&
run L_appnp_4_task_split --gpus 4 --workload L --model appnpstack --task-split on &&
run L_graphsage2_4 --gpus 4 --workload L --model graphsage2 &&
run L_gcn_3 --gpus 3 --workload L --model gcn &&
run L_gat_4 --gpus 4 --workload L --model gat
exit 0
