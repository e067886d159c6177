#!/usr/bin/env python3
"""One-off soak of tests/test_gpu_fuzz.py's differential fuzz beyond the seeds the suite pins (0..319): every seed in
[first, last) through run_case, failures collected (seed, kind, message) instead of stopping at the first.
Usage: python tools/fuzz_soak.py first last            (conv layers: run_case)
       python tools/fuzz_soak.py --models first last   (whole models: run_model_case)
       python tools/fuzz_soak.py --fused first last    (rgbx_fused_layer_f32 by option: run_fused_layer_case)
       python tools/fuzz_soak.py --ref64 seed [seed ...]   (the listed seeds against the oracle computing in float64)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fuzz as F
from conftest import usable_cpus

torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))  # the oracle half: the cgroup's CPUs, not the host's


def main():
    dev = torch.device("cuda")
    if sys.argv[1] == "--ref64":
        rc = 0
        for seed in map(int, sys.argv[2:]):
            for dtype in (torch.float32, torch.float64):
                try:
                    F.run_case(dev, seed, dtype)
                    print(f"seed {seed} vs the {dtype} oracle: within tolerance", flush=True)
                except AssertionError as exc:
                    print(f"seed {seed} vs the {dtype} oracle: {repr(exc)[:260]}", flush=True)
                    rc |= dtype == torch.float64
        return int(rc)
    which = sys.argv[1] if sys.argv[1].startswith("--") else ""
    if which:
        sys.argv.pop(1)
    run, kinds = {"--models": (F.run_model_case, F.MODEL_KINDS), "--fused": (F.run_fused_layer_case, ["fused_layer"]),
                  "": (F.run_case, F.KINDS)}[which]
    first, last = int(sys.argv[1]), int(sys.argv[2])
    bad, t0, mark = [], time.time(), time.time()
    for seed in range(first, last):
        try:
            run(dev, seed)
        except Exception as exc:  # noqa: BLE001 - collected, reported below
            bad.append((seed, kinds[seed % len(kinds)], repr(exc)[:300]))
            print("FAIL", bad[-1], flush=True)
        if time.time() - mark > 30:
            mark = time.time()
            print(f"seed {seed} ({seed - first + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s)", flush=True)
    torch.cuda.synchronize()
    print(f"soak [{first}, {last}): {last - first} cases, {len(bad)} failures in {time.time() - t0:.0f} s", flush=True)
    for b in bad:
        print(b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
