#!/bin/bash
# round 4, call 40: bench.py --gpus N over RCCL with the ranks sharing the one GPU, link rate of the cost model pinned to what an
# 8-GPU node is assumed to give (RGBX_LINK_GBS=60) so that the schemes a real node would run are the ones rehearsed
mkdir -p gpurun_out/r04
run() {  # name, then bench arguments
  name=$1; shift
  RGBX_LINK_GBS=60 RGBX_LINK_LATENCY_US=30 timeout -k 10 400 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r04/c40_$name.json 2> gpurun_out/r04/c40_$name.err
  echo "== $name rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c40_$name.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "error")})
    print("  ", d.get("config", {}).get("parallelism"))
    print("  ", d.get("launcher"), "| loss", d.get("final_train_loss"), d.get("loss_first_last"))
except Exception as e:
    print("no line:", e)
PY
  grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" gpurun_out/r04/c40_$name.err | tail -5 | cut -c1-300
}
run L_gcn_4 --gpus 4 --workload L --model gcn &&
run L_graphsage_4 --gpus 4 --workload L --model graphsage &&
run L_appnp_4 --gpus 4 --workload L --model appnpstack &&
run L_gcn_2 --gpus 2 --workload L --model gcn &&
run S_gat_4 --gpus 4 --workload S --model gat &&
run L_gcn_4_halo --gpus 4 --workload L --model gcn --exchange halo
exit 0
