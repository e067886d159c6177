#!/usr/bin/env python3
"""Stage by stage through one seed of the whole-model fuzz (conv stacks: gcn / graphsage / graphsage2): activations after every
conv and every training-mode BatchNorm, HIP (float32) and the oracle in float32, both against the oracle in float64; per stage
the smallest column standard deviation (BatchNorm divides by it: where it nears sqrt(eps) = 3e-3, rounding of the conv's output
is amplified by |x| / std). Usage: python tools/stack_stage_probe.py SEED"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fuzz as F
from oracle import ref_cpu as O

seed = int(sys.argv[1])
desc, model, ref_fn, ei, x, y, masks = F.make_model_case(seed)
print(desc)
kind = F.MODEL_KINDS[seed % len(F.MODEL_KINDS)]
L = model.num_layers
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}


def oracle_stages(dtype):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    conv = {"gcn": lambda i, v: O.gcn_conv(v, ei, p[f"convs.{i}.lin.weight"], p[f"convs.{i}.bias"]),
            "graphsage": lambda i, v: O.my_sage_conv(v, ei, p[f"convs.{i}.lin_l.weight"], p[f"convs.{i}.lin_l.bias"],
                                                     p[f"convs.{i}.lin_r.weight"], p[f"convs.{i}.lin_r.bias"]),
            "graphsage2": lambda i, v: O.sage_conv(v, ei, p[f"convs.{i}.lin_l.weight"], p[f"convs.{i}.lin_l.bias"],
                                                   p[f"convs.{i}.lin_r.weight"])}[kind]
    out, v = [], x.to(dtype)
    for i in range(L):
        v = conv(i, v)
        out.append((f"conv{i}", v))
        if i < L - 1:
            v = O.batch_norm(v, p, f"bns.{i}.", True)
            out.append((f"bn{i}", v))
    return out


dev = torch.device("cuda")
model.to(dev).train()
hip, v = [], x.to(dev)
with torch.no_grad():
    for i in range(L):
        v = model.convs[i](v, ei.to(dev))
        hip.append(v.cpu())
        if i < L - 1:
            v = model.bns[i](v)
            hip.append(v.cpu())
r64, r32 = oracle_stages(torch.float64), oracle_stages(torch.float32)
for a, (name, b64), (_, b32) in zip(hip, r64, r32):
    scale = b64.abs().max().item()
    std = b64.std(0, unbiased=False)
    print(f"{name:6s} |ref|max {scale:9.3e} column std min {std.min().item():9.3e} median {std.median().item():9.3e}   HIP vs f64 "
          f"{(a.double() - b64).abs().max().item() / scale:9.2e}   f32 oracle vs f64 {(b32.double() - b64).abs().max().item() / scale:9.2e}")
# the same stages with every conv fed the FLOAT64 run's input (rounded to float32): each conv's own error, no propagation
print("each conv on the float64 run's input:")
prev = x
with torch.no_grad():
    for i in range(L):
        got = model.convs[i](prev.float().to(dev), ei.to(dev)).cpu().double()
        ref = dict(r64)[f"conv{i}"]
        print(f"conv{i}  HIP vs f64 {(got - ref).abs().max().item() / ref.abs().max().item():9.2e}  (|row0 - row1| max of the reference "
              f"{(ref[0] - ref[-1]).abs().max().item():9.3e})")
        if i < L - 1:
            prev = dict(r64)[f"bn{i}"]
# the model's own routes: the training loss as the harness takes it (masked_ce: loss inside the last conv's kernel where a
# route offers it) and the module forward's logits, against the float64 oracle
from rgb_experiment_amd.graph import clear_cache
from rgb_experiment_amd.models._stack import masked_ce

nll = torch.nn.functional.nll_loss
clear_cache()
model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
model.train()
loss, stats = masked_ce(model, {"x": x.to(dev), "edge_index": ei.to(dev)}, y.to(dev), masks[0].to(dev))
model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
out = model(x.to(dev), ei.to(dev))
ref = ref_fn({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double(), True)
print(f"train mask {masks[0].tolist()} y {y.tolist()}")
print(f"masked_ce loss {loss.item():.9f}  rows {stats[1].item()}   module logits -> loss "
      f"{nll(out['out'][masks[0].to(dev)], y.to(dev)[masks[0].to(dev)]).item():.9f}   float64 oracle "
      f"{nll(ref['out'][masks[0]], y[masks[0]]).item():.9f}")
print("module logits", out["emb"].detach().cpu().tolist(), "\noracle logits", ref["emb"].tolist())
# which hand-over carries the difference: the same training forward with (B) no BatchNorm handed to the following conv,
# (C) additionally no column sums handed from the preceding conv's kernel to the BatchNorm
e64 = ref["emb"]
for label, no_after, no_sums in (("A as shipped", False, False), ("B no forward_after_bn", True, False),
                                 ("C nor column sums from the conv", True, True), ("D column sums off only", False, True)):
    clear_cache()
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    for c in model.convs:
        c.__dict__.pop("forward_after_bn", None)
        c.__dict__.pop("emits_colsums", None)
        if no_after:
            c.forward_after_bn = None
        if no_sums:
            c.emits_colsums = False
    got = model(x.to(dev), ei.to(dev))["emb"].detach().cpu().double()
    print(f"{label:34s} logits vs float64 {(got - e64).abs().max().item():9.2e}")
