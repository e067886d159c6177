#!/bin/bash
# round 4, call 10: two statistics sets from one eval forward: tests, then the default bench line
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tee $O/c10_tests.log | tail -4 || exit 1
python bench.py --steps 20 --warmup 5 > $O/c10_bench_default.json 2> $O/c10_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/c10_bench_default.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","median_ms_per_step","spmm_ms","epochs_per_s_identical_results"):
    print(k, d.get(k))
print({k:round(v["avg_ms"],3) for k,v in d["kernel_ms_by_variant"].items()}, "yardstick", round(d["yardstick"]["avg_ms"],3))
for k in ("identical_results_same_run","cached_input_aggregate_same_run","configs_1_same_run","configs_0_same_run"):
    b=d.get(k) or {}
    print(k, {a:b.get(a) for a in ("ms_per_step","median_ms_per_step","primary_ms_per_step","primary_ms_per_epoch","eager_ms_per_epoch","error")})
PY
