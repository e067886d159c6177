#!/usr/bin/env python3
"""Kernel-level A/B timing of the aggregation kernels on the benchmark graphs (interleaved rounds in one
process, HIP events on the launch stream). Usage: python tools/spmm_bench.py [S|L] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import WORKLOADS, spmm_alg_bytes, synth
from rgb_experiment_amd import ops
from rgb_experiment_amd.graph import LOOPS_ADD_REMAINING, get_graph


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "L"]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    N, E, d = wl["N"], wl["E"], wl["d"]
    dev = torch.device("cuda:0")
    ei, x, _ = synth(N, E, d)
    ei, x = ei.to(dev), x.to(dev)
    g = get_graph(ei, N, LOOPS_ADD_REMAINING)
    bias = torch.randn(d, device=dev)
    out = torch.empty_like(x)
    scratch = torch.empty(512 * 1024 * 1024 // 4, device=dev)
    variants = {
        "gcn_fwd": lambda: ops.spmm_raw(g.fwd, g.w, None, x, out=out),
        "gcn_fwd+bias": lambda: ops.spmm_raw(g.fwd, g.w, None, x, out=out, bias=bias),
        "gcn_bwd(transposed)": lambda: ops.spmm_raw(g.bwd, g.w_t, None, x, out=out),
        "sum_noweights": lambda: ops.spmm_raw(g.fwd, None, None, x, out=out),
        "gcn_fwd_after_2GB_stream": lambda: (scratch.add_(1.0), ops.spmm_raw(g.fwd, g.w, None, x, out=out)),
        "stream_2GB_alone": lambda: scratch.add_(1.0),
    }
    for fn in variants.values():
        fn()
    torch.cuda.synchronize()
    res = {k: [] for k in variants}
    for _ in range(rounds):
        for k, fn in variants.items():
            res[k].append(timed(fn, 10))
    alg = spmm_alg_bytes(N, g.fwd.nnz, d)
    for k, v in res.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"{k:28s} median {med:8.3f} ms  min {v[0]:8.3f}  -> {alg / med / 1e6:8.1f} GB/s algorithmic (if one SpMM)")


if __name__ == "__main__":
    main()
