#!/usr/bin/env python3
"""experiment() at a BASELINE size with its DEFAULT shortcuts (share_eval_forward, cache_input_aggregate="auto") against the
reference-shaped epoch (both off): the two runs must return the same history and weights bit for bit; prints both wall times.
Usage: python tools/exp_defaults_check.py [S|L] [model] [epochs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import rgb_experiment_amd as R
from bench import MODELS, WORKLOADS, synth


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "S"]
    name = sys.argv[2] if len(sys.argv) > 2 else "gcn"
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    ei, x, y = synth(wl["N"], wl["E"], wl["d"])
    data = R.Data(x=x, y=y, edge_index=ei)
    params = {k: v for k, v in MODELS[name][0].items()}
    runs = {}
    R.experiment(params, specify_data=True, data=data, model_name=name, learning_rate=0.01, epoch=1, need_to_reappear=True,
                 print_print=False, need_all_metrics=False)  # warm-up: library load, rocPRIM kernels, allocator
    for tag, kw in (("defaults", {}), ("reference-shaped", dict(share_eval_forward=False, cache_input_aggregate=False))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = R.experiment(params, specify_data=True, data=data, model_name=name, learning_rate=0.01, epoch=epochs,
                           need_to_reappear=True, print_print=False, return_model=True, need_all_metrics=False,
                           implement_early_stopping=False, **kw)
        torch.cuda.synchronize()
        runs[tag] = (res, time.perf_counter() - t0)
        print(f"{tag:17s} {runs[tag][1]:7.2f} s for {epochs} epochs incl. set-up; ACC {res['ACC']:.4f}; "
              f"last train / val / test loss {res['history']['train_loss'][-1]:.6f} {res['history']['val_loss'][-1]:.6f} "
              f"{res['history']['test_loss'][-1]:.6f}", flush=True)
    a, b = runs["defaults"][0], runs["reference-shaped"][0]
    same_hist = all(a["history"][k] == b["history"][k] for k in a["history"])
    same_w = all(torch.equal(p, q) for p, q in zip(a["model"].state_dict().values(), b["model"].state_dict().values()))
    print(f"histories identical: {same_hist}; weights identical: {same_w}; ACC identical: {a['ACC'] == b['ACC']}")
    if not (same_hist and same_w):
        sys.exit(1)


if __name__ == "__main__":
    main()
