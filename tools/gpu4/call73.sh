#!/bin/bash
# round 4, call 73: ingest fuzz (300 pinned cases) and a soak of 3,000 more
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_ingest.py -q -x -k "random_lists" --durations=2 2>&1 | tail -5 | cut -c1-300
timeout -k 10 600 python - <<'PY' 2>&1 | tee gpurun_out/r04/c73_ingest_soak.txt | tail -8 | cut -c1-300
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_ingest as T
dev = torch.device("cuda:0")
bad, t0, mark = [], time.time(), time.time()
for seed in range(300, 3300):
    try:
        T.ingest_fuzz_case(dev, seed)
    except Exception as exc:
        bad.append((seed, repr(exc)[:200])); print("FAIL", bad[-1], flush=True)
    if time.time() - mark > 30:
        mark = time.time(); print("seed", seed, len(bad), "failures", flush=True)
print(f"ingest soak [300, 3300): 3000 cases, {len(bad)} failures in {time.time() - t0:.0f} s")
PY
exit 0
