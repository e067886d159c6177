#!/usr/bin/env python3
"""What would keeping the gathered table in L2 buy at narrow widths? The row-gather kernel at d = 8 / 16 / 32 on graphs of the
benchmark's size (N = 2 M targets, E = 60 M) whose SOURCES are confined to a window of the node range: 4 MB of table (one XCD's
L2), 16 MB, 64 MB = the whole table at d = 8. The window runs are the upper bound of any scheme that slices the sources (one
pass per slice): same edges per target, same outputs written, only the gathered rows are closer together.
Usage: python tools/source_window.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rgb_experiment_amd import ops
from rgb_experiment_amd.graph import LOOPS_KEEP, clear_cache, get_graph

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, E = 2_000_000, 60_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
dst = torch.randint(0, N, (E,), generator=g, device=dev)
x = torch.randn(N, 32, generator=g, device=dev)
for d in (8, 16, 32):
    xd = x[:, :d].contiguous()
    for window_mb in (4, 16, 64, None):
        rows = N if window_mb is None else min(N, window_mb * (1 << 20) // (4 * d))
        src = torch.randint(0, rows, (E,), generator=g, device=dev)
        graph = get_graph(torch.stack([src, dst]), N, LOOPS_KEEP)
        out = torch.empty_like(xd)
        ops.spmm_raw(graph.fwd, None, None, xd, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.spmm_raw(graph.fwd, None, None, xd, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print(f"d={d:3d}  sources within {('%d MB' % window_mb) if window_mb else 'all N'} of table ({rows} rows): {ms:6.3f} ms "
              f"({E / ms / 1e6:6.1f} G edges/s)", flush=True)
        del graph
        clear_cache()
