#!/bin/bash
# round 4, call 39: tests/test_gpu_dist.py with its backend probe (expected: RCCL), then bench.py --gpus 2 / 4 on the one GPU
# (RCCL, ranks sharing the device: the product's multi-rank path end to end), workload S
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -q --durations=6 -x -rs 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$" > gpurun_out/r04/c39_gpu_dist.log
tail -16 gpurun_out/r04/c39_gpu_dist.log | cut -c1-300
for n in 2 4; do
  timeout -k 10 420 python bench.py --gpus $n --workload S --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04/c39_bench_S_rccl_shared_$n.json 2> gpurun_out/r04/c39_bench_S_rccl_shared_$n.err
  echo "bench --gpus $n rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c39_bench_S_rccl_shared_$n.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("metric", "value", "n_gpus", "ms_per_step", "error")})
    print(d.get("config", {}).get("parallelism"))
    print(d.get("ranks_share_devices", {}).get("ranks"), d.get("link_gbs_measured"), d.get("launcher"))
except Exception as e:
    print("no line:", e)
PY
  grep -v "alt_rsmi\|LL cutoff\|^$" gpurun_out/r04/c39_bench_S_rccl_shared_$n.err | tail -8 | cut -c1-300
done
exit 0
