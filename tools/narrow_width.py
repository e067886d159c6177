#!/usr/bin/env python3
"""The GCN SpMM of the benchmark graph at narrow widths (what a C = 7 last layer, or a column shard of a partitioned
run, aggregates at): ms per launch for d = 8 .. 128, optionally on a row-group share of the graph. Run it under
`rocprofv3 --pmc ...` to attribute the counters per width (the kernel's template arguments differ per width).
Usage: python tools/narrow_width.py [S|L] [reps] [widths, comma separated]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import WORKLOADS, spmm_alg_bytes, synth
from rgb_experiment_amd import ops
from rgb_experiment_amd.graph import LOOPS_ADD_REMAINING, get_graph


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "L"]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    widths = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,16,32,64,128").split(",")]
    N, E = wl["N"], wl["E"]
    dev = torch.device("cuda:0")
    ei, x, _ = synth(N, E, 128)
    g = get_graph(ei.to(dev), N, LOOPS_ADD_REMAINING)
    _ = g.w
    x = x.to(dev)
    for d in widths:
        xd = x[:, :d].contiguous()
        out = torch.empty_like(xd)
        ops.spmm_raw(g.fwd, g.w, None, xd, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.spmm_raw(g.fwd, g.w, None, xd, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        alg = spmm_alg_bytes(N, g.fwd.nnz, d)
        lines = g.fwd.nnz * max(1, (4 * d + 127) // 128)
        print(f"d={d:4d}  {ms:7.3f} ms  algorithmic {alg / ms / 1e6:8.1f} GB/s  gathered 128-B lines {lines / ms / 1e6:7.2f} G/s "
              f"= {lines * 128 / ms / 1e6:8.1f} GB/s of lines", flush=True)


if __name__ == "__main__":
    main()
