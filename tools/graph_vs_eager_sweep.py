#!/usr/bin/env python3
"""One-off sweep: experiment() with the epoch captured and replayed as a hipGraph against the eager loop, over random model
families / depths / widths / class counts / graph shapes (tests/test_gpu_fuzz.make_model_case's generator): same histories, same
trained weights, same accuracy — and every other default of experiment() at work (kept input aggregate, one eval forward for
both masks). Usage: python tools/graph_vs_eager_sweep.py first last"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random

import numpy as np
import torch

import rgb_experiment_amd as R
import test_gpu_fuzz as F


def case(seed):
    rng = random.Random(seed)
    name = ["gcn", "graphsage", "graphsage2", "gat", "appnpstack", "sgc", "gin", "dagnn"][seed % 8]
    n = rng.choice([33, 100, 257, 700, 3000])
    f = rng.choice([3, 8, 32, 33, 64, 128])
    c = rng.choice([2, 7, 32, 40, 128, 130] if n >= 3000 else [2, 7, 32] if n >= 700 else [2, 5])  # (every class must reach the train part)
    ei = F.make_graph(rng, n)
    g = torch.Generator().manual_seed(seed)
    data = R.Data(x=torch.randn(n, f, generator=g), y=torch.randint(0, c, (n,), generator=g), edge_index=ei)
    params = R.InitialParameters.defaults_for(name)
    if "hidden_unit" in params:
        params["hidden_unit"] = rng.choice([8, 16, 32, 64, 100, 128]) if name != "gat" else rng.choice([4, 8, 16])
    if "num_layers" in params:
        params["num_layers"] = rng.choice([2, 2, 3])
    if "dropout_rate" in params:
        params["dropout_rate"] = 0.0
    desc = f"seed={seed} {name} n={n} E={ei.size(1)} f={f} classes={c} params={params}"
    runs = []
    for graphed in (False, "always"):
        runs.append(R.experiment(dict(params), specify_data=True, data=data, model_name=name, learning_rate=0.01, epoch=5,
                                 need_to_reappear=True, print_print=False, return_model=True, use_hip_graph=graphed,
                                 need_all_metrics=False))
    a, b = runs
    assert b["used_hip_graph"] and not a["used_hip_graph"], (desc, "no capture")
    for key in ("train_loss", "val_loss", "test_loss", "train_acc", "val_acc", "test_acc"):
        assert np.allclose(a["history"][key], b["history"][key], rtol=0, atol=5e-6, equal_nan=True), (desc, key, a["history"][key], b["history"][key])
    for (ka, va), (kb, vb) in zip(a["model"].state_dict().items(), b["model"].state_dict().items()):
        assert ka == kb and torch.allclose(va.float(), vb.float(), atol=2e-6, equal_nan=True), (desc, ka)
    assert abs(a["ACC"] - b["ACC"]) < 1e-9, (desc, "ACC")


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    bad, t0, mark = [], time.time(), time.time()
    import faulthandler
    for seed in range(first, last):
        faulthandler.dump_traceback_later(90, exit=True)  # a case that hangs ends the sweep with its stack
        print(f"seed {seed} ...", flush=True)
        try:
            case(seed)
        except ValueError as exc:
            if "never end" not in str(exc):  # (more classes than training rows: utils/mask.get_whole_mask refuses the split)
                raise
            print(f"seed {seed}: split refused ({str(exc)[:60]}...)", flush=True)
        except Exception as exc:  # noqa: BLE001
            bad.append((seed, repr(exc)[:500]))
            print("FAIL", bad[-1], flush=True)
        if time.time() - mark > 20:
            mark = time.time()
            print(f"seed {seed} ({len(bad)} failures, {time.time() - t0:.0f} s)", flush=True)
    print(f"graph vs eager [{first}, {last}): {last - first} cases, {len(bad)} failures in {time.time() - t0:.0f} s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
