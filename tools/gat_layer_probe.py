#!/usr/bin/env python3
"""Stage by stage through the whole-model fuzz's seed 2528 (4-layer GAT on a hub-dominated 100-node graph): the activations
after every conv and every training-mode BatchNorm, HIP (float32) and the oracle in float32, both against the oracle in
float64 — which stage loses the accuracy the first layers' gradients then lack?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fuzz as F
from oracle import ref_cpu as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2528
desc, model, ref_fn, ei, x, y, masks = F.make_model_case(seed)
print(desc)
L, heads = model.num_layers, model.heads
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}


def oracle_stages(dtype):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    out, v = [], x.to(dtype)
    for i in range(L):
        last = i == L - 1
        v = O.gat_conv(v, ei, p[f"convs.{i}.lin_src.weight"], p[f"convs.{i}.att_src"], p[f"convs.{i}.att_dst"],
                       p[f"convs.{i}.bias"], 1 if last else heads, concat=not last)
        out.append((f"conv{i}", v))
        if not last:
            v = O.batch_norm(v, p, f"bns.{i}.", True)
            out.append((f"bn{i}", v))
    return out


dev = torch.device("cuda")
model.to(dev).train()
hip, v = [], x.to(dev)
with torch.no_grad():
    for i in range(L):
        v = model.convs[i](v, ei.to(dev))
        hip.append((f"conv{i}", v.cpu()))
        if i < L - 1:
            v = model.bns[i](v)
            hip.append((f"bn{i}", v.cpu()))
r64, r32 = oracle_stages(torch.float64), oracle_stages(torch.float32)
for (name, a), (_, b64), (_, b32) in zip(hip, r64, r32):
    scale = b64.abs().max().item()
    colstd = b64.std(0).min().item()
    print(f"{name:6s} |ref|max {scale:9.3e} min column std {colstd:9.3e}   HIP vs f64 {(a.double() - b64).abs().max().item() / scale:9.2e}"
          f"   f32 oracle vs f64 {(b32.double() - b64).abs().max().item() / scale:9.2e}")

# ---- backward: gradient w.r.t. every stage's output, HIP and the float32 oracle against the float64 oracle -----------------
nll = torch.nn.functional.nll_loss


def oracle_grads(dtype):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    p = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    stages, v = [], x.to(dtype)
    for i in range(L):
        last = i == L - 1
        v = O.gat_conv(v, ei, p[f"convs.{i}.lin_src.weight"], p[f"convs.{i}.att_src"], p[f"convs.{i}.att_dst"],
                       p[f"convs.{i}.bias"], 1 if last else heads, concat=not last)
        v.retain_grad()
        stages.append((f"conv{i}", v))
        if not last:
            v = O.batch_norm(v, p, f"bns.{i}.", True)
            v.retain_grad()
            stages.append((f"bn{i}", v))
    nll(torch.log_softmax(v, 1)[masks[0]], y[masks[0]]).backward()
    return [(n_, t.grad.detach()) for n_, t in stages], {k: v.grad for k, v in p.items() if v.grad is not None}


model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
model.train()
stages, v = [], x.to(dev)
for i in range(L):
    v = model.convs[i](v, ei.to(dev))
    v.retain_grad()
    stages.append((f"conv{i}", v))
    if i < L - 1:
        v = model.bns[i](v)
        v.retain_grad()
        stages.append((f"bn{i}", v))
nll(torch.log_softmax(v, 1)[masks[0].to(dev)], y.to(dev)[masks[0].to(dev)]).backward()
(g64, p64), (g32, p32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
print("gradient w.r.t. the stage's output (max |diff| / max |ref|), and its column sums:")
for (name, t), (_, a64), (_, a32) in zip(stages, g64, g32):
    a = t.grad.detach().cpu().double()
    sc = a64.abs().max().item()
    cs64 = a64.sum(0)
    csc = max(cs64.abs().max().item(), 1e-30)
    print(f"{name:6s} |g|max {sc:9.3e}  HIP {(a - a64).abs().max().item() / sc:9.2e}  f32 oracle {(a32.double() - a64).abs().max().item() / sc:9.2e}"
          f"   column sums |max| {csc:9.3e}: HIP {(a.sum(0) - cs64).abs().max().item() / csc:9.2e}  f32 oracle "
          f"{(a32.double().sum(0) - cs64).abs().max().item() / csc:9.2e}")
print("parameters:")
for k, prm in model.named_parameters():
    if k in p64 and prm.grad is not None:
        r = p64[k]
        sc = max(r.abs().max().item(), 1e-30)
        print(f"  {k:28s} |g|max {sc:9.3e}  HIP {(prm.grad.cpu().double() - r).abs().max().item() / sc:9.2e}  f32 oracle "
              f"{(p32[k].double() - r).abs().max().item() / sc:9.2e}")

# ---- one layer on its own: layer i's input and output gradient from the float64 run, layer i's backward on HIP ------------
from rgb_experiment_amd import nn as RN

p64_sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
stage64 = dict(oracle_stages(torch.float64))
gout64 = dict(g64)
for i in (1, 2):
    xin = stage64[f"bn{i - 1}"].float()
    gout = gout64[f"conv{i}"].float()
    W, a_s, a_d, b = (sd[f"convs.{i}.lin_src.weight"], sd[f"convs.{i}.att_src"], sd[f"convs.{i}.att_dst"], sd[f"convs.{i}.bias"])
    # float64 reference of this layer alone (same float32 inputs)
    xr = xin.double().requires_grad_(True)
    pr = [t.double().clone().requires_grad_(True) for t in (W, a_s, a_d, b)]
    O.gat_conv(xr, ei, pr[0], pr[1], pr[2], pr[3], heads, concat=True).backward(gout.double())
    conv = RN.GATConv(W.size(1), W.size(0) // heads, heads).to(dev)
    with torch.no_grad():
        conv.lin_src.weight.copy_(W.to(dev)); conv.att_src.copy_(a_s.to(dev)); conv.att_dst.copy_(a_d.to(dev)); conv.bias.copy_(b.to(dev))
    xd = xin.to(dev).requires_grad_(True)
    conv(xd, ei.to(dev)).backward(gout.to(dev))
    rel = lambda a, r: (a.detach().cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
    print(f"layer {i} alone (float64 inputs cast to float32): g_x {rel(xd.grad, xr.grad):.2e}  g_W {rel(conv.lin_src.weight.grad, pr[0].grad):.2e}"
          f"  g_att_src {rel(conv.att_src.grad, pr[1].grad):.2e}  g_att_dst {rel(conv.att_dst.grad, pr[2].grad):.2e}"
          f"   |x|max {xin.abs().max():.2f} |gout|max {gout.abs().max():.2e}")
    # the score range of this layer: how saturated is the softmax
    h = (xin.double() @ W.double().t()).view(-1, heads, W.size(0) // heads)
    s_src = (h * a_s.double().view(1, heads, -1)).sum(-1)
    s_dst = (h * a_d.double().view(1, heads, -1)).sum(-1)
    print(f"          scores: a_src in [{s_src.min():.1f}, {s_src.max():.1f}], a_dst in [{s_dst.min():.1f}, {s_dst.max():.1f}]")

# ---- what in HIP's gradient w.r.t. conv1's output does layer 1's backward amplify? ----------------------------------------
hip_g = {n_: t.grad.detach().cpu() for n_, t in stages}
i = 1
e = (hip_g[f"conv{i}"].double() - gout64[f"conv{i}"])          # HIP's error in the gradient that enters layer 1's backward
print(f"error entering layer {i}'s backward: max {e.abs().max():.2e} (|g|max {gout64[f'conv{i}'].abs().max():.2e}); "
      f"largest rows {e.abs().max(1)[0].topk(4)}; row 2-norms top {e.norm(dim=1).topk(4)[0].tolist()}")
deg_in = torch.bincount(ei[1], minlength=x.size(0))
deg_out = torch.bincount(ei[0], minlength=x.size(0))
print(f"hub target {int(deg_in.argmax())} ({int(deg_in.max())} in-edges), hub source {int(deg_out.argmax())} ({int(deg_out.max())} out-edges)")
xin = stage64[f"bn{i - 1}"]
W, a_s, a_d, b = (p64_sd[f"convs.{i}.lin_src.weight"], p64_sd[f"convs.{i}.att_src"], p64_sd[f"convs.{i}.att_dst"], p64_sd[f"convs.{i}.bias"])


def jac(gvec):
    xr = xin.clone().requires_grad_(True)
    O.gat_conv(xr, ei, W, a_s, a_d, b, heads, concat=True).backward(gvec)
    return xr.grad


je = jac(e)
print(f"layer {i}'s backward applied to that error (float64): max {je.abs().max():.2e} — vs HIP's error in the gradient w.r.t. "
      f"layer {i}'s input {(hip_g[f'bn{i - 1}'].double() - gout64[f'bn{i - 1}']).abs().max():.2e}")
for row in e.abs().max(1)[0].topk(3)[1].tolist():
    one = torch.zeros_like(e)
    one[row] = e[row]
    print(f"   the error's row {row} alone (in-degree {int(deg_in[row])}, out-degree {int(deg_out[row])}): |row| {e[row].abs().max():.2e} -> {jac(one).abs().max():.2e}")

# ---- is HIP's layer-1 backward accurate AT ITS OWN inputs? (float64 backward evaluated at HIP's x1 and HIP's gout1) ----------
hip_x = {n_: t.detach().cpu() for n_, t in stages}
xr = hip_x["bn0"].double().requires_grad_(True)
O.gat_conv(xr, ei, W, a_s, a_d, b, heads, concat=True).backward(hip_g["conv1"].double())
own = (hip_g["bn0"].double() - xr.grad).abs().max().item()
print(f"layer 1's backward, HIP vs float64 evaluated at HIP's OWN input and output gradient: {own:.2e} "
      f"(against float64's own chain: {(hip_g['bn0'].double() - gout64['bn0']).abs().max():.2e})")
# how far do the two chains' pre-activation scores differ, and how close to the LeakyReLU kink does the closest one sit?
def scores(xv):
    h = (xv.double() @ W.t()).view(-1, heads, W.size(0) // heads)
    rei, _ = O.remove_self_loops(ei)
    rei, _ = O.add_self_loops(rei, num_nodes=xv.size(0))
    return (h * a_s.view(1, heads, -1)).sum(-1)[rei[0]] + (h * a_d.view(1, heads, -1)).sum(-1)[rei[1]]
s_hip, s_64 = scores(hip_x["bn0"]), scores(stage64["bn0"])
flipped = ((s_hip > 0) != (s_64 > 0)).sum().item()
print(f"pre-activation scores: closest to the kink |s| = {s_64.abs().min():.2e}, chains differ by at most {(s_hip - s_64).abs().max():.2e}, "
      f"{flipped} of {s_64.numel()} on different sides of the kink")
