#!/usr/bin/env python3
"""Conv-layer fuzz seed 34429 (GIN, n = 700, in = 256): eps.grad 0.19 off against the float64 oracle. Is it a ReLU kink — a
pre-activation of nn's first Linear so close to zero that float32 rounding decides its side, where d eps picks up
g * (x W^T) for that entry or not? Rebuilds the case as tests/test_gpu_fuzz.run_case does, prints the smallest |pre-activation|
of the float64 run, HIP's value there, and eps.grad of HIP / float64 / float64 with that one derivative flipped.
Usage: python tools/gin_kink_probe.py [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_fuzz as F
from oracle import ref_cpu as O
from rgb_experiment_amd import nn as RN

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 34429
rng = random.Random(seed)
assert F.KINDS[seed % len(F.KINDS)] == "gin"
n = rng.choice([1, 2, 3, 31, 32, 33, 64, 100, 257, 700])
f_in, f_out = rng.choice(F.WIDTHS), rng.choice(F.WIDTHS)
ei = F.make_graph(rng, n)
g = torch.Generator().manual_seed(seed)
x = torch.randn(n, f_in, generator=g)
rng.random(), rng.random()
torch.manual_seed(seed)
hid = rng.choice([8, 32, 64])
conv = RN.GINConv(torch.nn.Sequential(RN.Linear(f_in, hid), torch.nn.ReLU(), RN.Linear(hid, f_out)), train_eps=True)
with torch.no_grad():
    conv.eps.fill_(rng.choice([0.0, 0.25, -0.5]))
    for p in conv.parameters():
        if p.dim() == 1 and p.numel() > 1:
            p.uniform_(-0.5, 0.5)
print(f"seed={seed} n={n} E={ei.size(1)} in={f_in} hidden={hid} out={f_out} eps={conv.eps.item()}")
sd = {k: v.detach().clone().double().requires_grad_(True) for k, v in conv.state_dict().items()}
xc = x.double()
agg = O.propagate(ei, xc, n, None, "add") + (1 + sd["eps"]) * xc
pre = agg @ sd["nn.0.weight"].t() + sd["nn.0.bias"]
ref = torch.relu(pre) @ sd["nn.2.weight"].t() + sd["nn.2.bias"]
go = torch.randn(ref.shape, generator=g)
ref.backward(go.double())
dev = torch.device("cuda")
conv.to(dev)
seen = {}


def keep_input(_module, inputs, _output):
    seen["pre"] = inputs[0].detach().cpu().double()  # (returns None: the module's output stays)


conv.nn[1].register_forward_hook(keep_input)
got = conv(x.to(dev), ei.to(dev))
got.backward(go.to(dev))
flat = pre.detach().abs().flatten()
order = torch.argsort(flat)[:5]
print("smallest |pre-activation| of the float64 run, and HIP's value there:")
for k in order.tolist():
    i, c = divmod(k, hid)
    print(f"  row {i} col {c}: float64 {pre[i, c].item(): .3e}   HIP {seen['pre'][i, c].item(): .3e}")
print(f"max |pre HIP - pre float64| = {(seen['pre'] - pre.detach()).abs().max().item():.2e} (scale {pre.detach().abs().max().item():.2f})")
i, c = divmod(order[0].item(), hid)
# d eps = sum_{i,c} relu'(pre) * (go W2)[i,c] * (x W0^T)[i,c]
g_h = (go.double() @ sd["nn.2.weight"].detach())
xw = xc @ sd["nn.0.weight"].detach().t()
term = (g_h[i, c] * xw[i, c]).item()
print(f"eps.grad: HIP {conv.eps.grad.item():.6f}   float64 {sd['eps'].grad.item():.6f}   difference {conv.eps.grad.item() - sd['eps'].grad.item():+.6f}")
print(f"the entry nearest the kink contributes g * (x W^T) = {term:+.6f} when its derivative is 1")
