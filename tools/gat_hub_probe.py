#!/usr/bin/env python3
"""Where does the whole-model fuzz's seed 2528 (4-layer GAT, 100 nodes, a 3,600-in-edge target and an 1,800-out-edge source)
lose its first-layer gradient accuracy? Per-parameter error against the float64 oracle under three settings of the row-split
threshold (hub rows chunked / whole) and with the hub-target pass on / off."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fuzz as F
from oracle import large as OL
from rgb_experiment_amd import graph as G
from rgb_experiment_amd import ops
from rgb_experiment_amd.graph import clear_cache
from rgb_experiment_amd.models._stack import masked_ce


def run(seed, threshold, hubs_pass):
    G.LONG_ROW_SLOTS = threshold
    clear_cache()
    desc, model, ref_fn, ei, x, y, masks = F.make_model_case(seed)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    out = ref_fn(sd, x.double(), True)["out"]
    torch.nn.functional.nll_loss(out[masks[0]], y[masks[0]]).backward()
    dev = torch.device("cuda")
    model.to(dev).train()
    real = ops._lib.load().rgbx_gat_bwd_dst_hubs_f32
    if not hubs_pass:
        ops._lib.load().rgbx_gat_bwd_dst_hubs_f32 = lambda *a: 0
    try:
        loss, _ = masked_ce(model, {"x": x.to(dev), "edge_index": ei.to(dev)}, y.to(dev), masks[0].to(dev))
        loss.backward()
    finally:
        ops._lib.load().rgbx_gat_bwd_dst_hubs_f32 = real
    got = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    rep = OL.compare_grads(got, {k: sd[k].grad.float() for k in got})
    worst = sorted(rep["per_param"].items(), key=lambda kv: -kv[1]["rel"])[:4]
    print(f"{desc}\n  threshold {threshold}, hub-target pass {hubs_pass}: max_rel {rep['max_rel']:.2e}; worst: " +
          ", ".join(f"{k} {v['rel']:.1e}" for k, v in worst), flush=True)


for seed in map(int, sys.argv[1:] or ["2528"]):
    for threshold, hubs in ((1024, True), (1024, False), (1 << 30, True), (256, True)):
        run(seed, threshold, hubs)
