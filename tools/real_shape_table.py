#!/usr/bin/env python3
"""Print the real-shape block(s) of a bench.py line (`--real-shape` output or the default line's real_shape_same_run)."""
import json
import sys


def show(tag, block):
    for name, m in block["models"].items():
        print(f"{tag} {name}: F={block['features']} ({block.get('feature_values', 'dense')}, over non-zeros: "
              f"{block.get('features_multiplied_over_nonzeros', False)}) C={block['classes']} peak {m['hbm_allocated_peak_gb']:.2f} GB")
        for lab in ("reference_epoch", "identical_results_epoch"):
            e = m[lab]
            print(f"  {lab}: {e['ms_per_epoch']:.2f} ms/epoch (timed launches {e['timed_launches_ms_per_epoch']:.2f})")
            for k, v in e["launches"].items():
                print(f"     {k:34s} n={v['n_per_epoch']:.1f} avg={v['avg_ms']:.3f} ms  per epoch {v['ms_per_epoch']:.3f}")
        for k, v in m["row_gather_roofline"].items():
            print(f"  roofline {k:30s} d={v['width']:3d} {v['achieved_gbs']:7.0f} GB/s = {v['frac_of_8TBs']:.3f} of 8 TB/s")


for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        if "real_shape" in r:
            show(path, r["real_shape"])
        for key, block in (r.get("real_shape_same_run") or {}).items():
            if isinstance(block, dict) and "models" in block:
                show(f"{path}:{key}", block)
