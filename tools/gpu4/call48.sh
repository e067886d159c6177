#!/bin/bash
# round 4, call 48: device sharing told from published PCI identities (store exchange before the communicator): bench.py --gpus 2/4
# under torch.distributed.run's agent store, experiment() as several ranks, and the rest of tests/test_gpu_dist.py
mkdir -p gpurun_out/r04
for n in 2 4; do
  RGBX_LINK_GBS=60 timeout -k 10 400 python bench.py --gpus $n --workload S --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04/c48_S_$n.json 2> gpurun_out/r04/c48_S_$n.err
  echo "bench --gpus $n rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04/c48_S_$n.json").read().strip().splitlines()[-1])
    print(d.get("metric")[:60], d.get("ms_per_step"), d.get("final_losses"), d.get("scheme"), d.get("ranks_share_devices", {}).get("ranks"), d.get("launcher", {}).get("attempt"), d.get("error"))
except Exception as e:
    print("no line:", e)
PY
  grep -v "alt_rsmi\|LL cutoff\|^$\|amdgpu.ids\|socket.cpp" gpurun_out/r04/c48_S_$n.err | tail -4 | cut -c1-300
done
timeout -k 10 800 python -m pytest tests/test_gpu_dist.py -q --durations=5 -rs -x 2>&1 | grep -v "alt_rsmi\|LL cutoff\|^$" > gpurun_out/r04/c48_gpu_dist.log
tail -12 gpurun_out/r04/c48_gpu_dist.log | cut -c1-300
exit 0
