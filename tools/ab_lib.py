#!/usr/bin/env python3
"""A/B of two builds of librgbx_hip.so on the benchmark graph: the same op closures run interleaved through build A
(the in-tree library) and build B (a variant built elsewhere, e.g. scratch/<name>/librgbx_hip.so), HIP events.
Usage: python tools/ab_lib.py <path to build B> [S|L] [rounds] [ops: gat,gcn,sage,spmm,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import WORKLOADS, synth
from rgb_experiment_amd import _lib, nn as RN, ops
from rgb_experiment_amd.graph import get_graph


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    path_b = sys.argv[1]
    wl = WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "L"]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    which = (sys.argv[4] if len(sys.argv) > 4 else "gat,gcn").split(",")
    N, E, d = wl["N"], wl["E"], wl["d"]
    dev = torch.device("cuda:0")
    ei, x, _ = synth(N, E, d)
    ei, x = ei.to(dev), x.to(dev)
    gy = torch.randn_like(x)
    lib_a, lib_b = _lib.load(), _lib.bind(path_b, strict=False)
    cases = {}
    if "gat" in which:
        for H, C in ((8, 16), (1, 128)):
            torch.manual_seed(0)
            conv = RN.GATConv(d, C, H, concat=H > 1).to(dev)
            h = x.clone().requires_grad_(True)

            def infer(conv=conv):
                with torch.no_grad():
                    conv(x, ei)

            def train(conv=conv, h=h):
                out = conv(h, ei)
                out.backward(gy[:, :out.size(1)])
            cases[f"gat H={H} C={C} inference forward"] = infer
            cases[f"gat H={H} C={C} train forward+backward"] = train
    if "gcn" in which or "sage" in which:
        torch.manual_seed(0)
        layers = []
        if "gcn" in which:
            layers.append(("gcn", RN.GCNConv(d, d).to(dev)))
        if "sage" in which:
            layers.append(("sage", RN.SAGEConv(d, d).to(dev)))
        for name, conv in layers:
            h = x.clone().requires_grad_(True)

            def infer(conv=conv):
                with torch.no_grad():
                    conv(x, ei)

            def train(conv=conv, h=h):
                conv(h, ei).backward(gy)
            cases[f"{name} inference forward"] = infer
            cases[f"{name} train forward+backward"] = train
    if "spmm" in which:
        g = get_graph(ei, N, 1)
        out = torch.empty_like(x)
        cases["gcn spmm"] = lambda: ops.spmm_raw(g.fwd, g.w, None, x, out=out)
    if "appnp" in which:  # the K-loop: every step gathers the table the previous step wrote
        g = get_graph(ei, N, 1)
        cases["appnp K=10 forward"] = lambda: ops.appnp_raw(g.fwd, g.w, x, 10, 0.1)
    res = {k: {"A": [], "B": []} for k in cases}
    for tag, lib in (("A", lib_a), ("B", lib_b)):  # warm-up both (graph build, allocator)
        _lib.use(lib)
        for fn in cases.values():
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in cases.items():
            for tag, lib in (("A", lib_a), ("B", lib_b)):
                _lib.use(lib)
                res[k][tag].append(timed(fn, 5))
    _lib.use(lib_a)
    for k, r in res.items():
        a, b = sorted(r["A"])[len(r["A"]) // 2], sorted(r["B"])[len(r["B"]) // 2]
        print(f"{k:42s} A {a:8.3f} ms   B {b:8.3f} ms   B/A {b / a:6.3f}", flush=True)


if __name__ == "__main__":
    main()
