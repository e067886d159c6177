#!/bin/bash
# round 4, call 53: the default line on the code as it stands (ops._MadeOn on the cached operands, dist/sharing in bench.py)
mkdir -p gpurun_out/r04
timeout -k 10 900 python bench.py > gpurun_out/r04/c53_bench_default.json 2> gpurun_out/r04/c53_bench_default.err
echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/c53_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "median_ms_per_step", "n_gpus")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic")})
print("yardstick", d.get("yardstick", {}).get("avg_ms"), "identical", d.get("identical_results_same_run", {}).get("ms_per_step"), d.get("epochs_per_s_identical_results"))
print("cpu", d.get("cpu_baseline"))
print("parity", json.dumps(d.get("parity"))[:400])
PY
tail -3 gpurun_out/r04/c53_bench_default.err | cut -c1-200
exit 0
