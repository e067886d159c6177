#!/bin/bash
# round 4, call 3: default bench line with the new fields (variants, yardstick, ingest, gradient parity at S, identical-results epoch)
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
( time python bench.py --steps 20 --warmup 5 ) > $O/c3_bench_default.json 2> $O/c3_bench_default.err; echo "bench rc=$?"
tail -3 $O/c3_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/c3_bench_default.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","median_ms_per_step","epochs_per_s","spmm_ms","kernel_ms_by_kind","kernel_ms_by_variant","yardstick","ingest_ms","epochs_per_s_identical_results","parity_gradients_at_S"):
    print(k, json.dumps(d.get(k)))
print("roofline", json.dumps({k:v for k,v in d["roofline"].items() if k!="note"}))
for k in ("identical_results_same_run","cached_input_aggregate_same_run","undirected_same_run"):
    b=d.get(k) or {}
    print(k, json.dumps({a:b.get(a) for a in ("ms_per_step","median_ms_per_step","epochs_per_s","per_step_ms","ingest_ms","error","aggregations_per_step")}))
PY
