#!/bin/bash
# round 4, call 8b: contended emulation with the per-step list
mkdir -p gpurun_out/r04
python bench.py --emulate-rank 8 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r04/c8b.json 2> gpurun_out/r04/c8b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/c8b.json").read().strip().splitlines()[-1])
c=d["emulated"]["contended"]
print(c.get("error"))
print(c["per_step_ms"], c["median_ms_per_step"], c["compute_only_median_ms_per_step"], c["exposed_exchange_wait_ms_per_step"])
print(d["per_rank"][0]["host_enqueue_ms_per_step"])
PY
