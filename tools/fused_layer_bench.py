#!/usr/bin/env python3
"""Microbenchmark of rgbx_fused_layer_f32 forms on rank 0's share of workload L at P = 8 (n_local = 250 k rows, the
[local; halo] CSR of the resident layer): aggregating launch with row-major vs blocked outputs, and the DENSE (return
stage) launch in its training / eval / backward forms against hipBLASLt. HIP events, interleaved rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import WORKLOADS, synth
from rgb_experiment_amd import ops
from rgb_experiment_amd.dist.plan import PartitionPlan
from rgb_experiment_amd.dist.graph import HipAggregator


def timed(fn, reps=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "L"]
    N, E, d = wl["N"], wl["E"], wl["d"]
    P = 8
    dev = torch.device("cuda:0")
    ei, x, y = synth(N, E, d)
    plan = PartitionPlan(ei.to(dev), N, P, 0, 1, "gcn")
    f = plan.fwd
    be = HipAggregator()
    agg = torch.cat([f.loc_agg, f.rem_agg])
    gather = torch.cat([f.loc_gather, f.n_local + f.rem_gather])
    csr, ws = be.prepare(agg, gather, f.n_local, torch.cat([f.loc_w, f.rem_w]))
    n_loc, n_ext = f.n_local, f.n_local + f.n_halo
    x_ext = torch.randn(n_ext, d, device=dev)
    W = torch.randn(d, d, device=dev) / d ** 0.5
    wt = W.t().contiguous()
    b = torch.randn(d, device=dev)
    C = 4
    blk = torch.empty((C, n_loc, d // C), device=dev)
    u = torch.randn((C, n_loc, d // C), device=dev)
    rows = torch.randn(n_loc, d, device=dev)
    s_, t_ = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
    rowsum = torch.rand(n_loc, device=dev)
    yl = y[:n_loc].to(dev)
    mask = (torch.arange(n_loc, device=dev) % 5) < 3
    scale = ops.mask_scale(yl, mask, d)
    cases = {
        "aggregate: row-major out (plain kernel)": lambda: ops.fused_layer(x_ext, wt, csr=csr, w=ws, bias=b),
        "aggregate: blocked out only (eval producer)": lambda: ops.fused_layer(x_ext, wt, csr=csr, w=ws, bias=b, want_out=False, out_blocked=blk),
        "aggregate: rows + blocked + z + colsums (training producer)": lambda: ops.fused_layer(x_ext, wt, csr=csr, w=ws, bias=b, out_blocked=blk, want_z=True, want_colsums=True),
        "aggregate: rows + z + colsums (plain kernel)": lambda: ops.fused_layer(x_ext, wt, csr=csr, w=ws, bias=b, want_z=True, want_colsums=True),
        "plain SpMM over the same CSR": lambda: ops.spmm_raw(csr, ws, None, x_ext),
        "dense: blocked in, pre-affine, z, loss gradient (training return stage)": lambda: ops.fused_layer(u, wt, bias=b, pre=(s_, t_, rowsum), want_z=True, ce=(yl, mask, scale)),
        "dense: blocked in, loss statistics only (eval return stage)": lambda: ops.fused_layer(u, wt, bias=b, ce=(yl, mask, None)),
        "dense: rows in, blocked out only (q = dy W)": lambda: ops.fused_layer(rows, wt, want_out=False, out_blocked=blk),
        "dense: rows in, rows out": lambda: ops.fused_layer(rows, wt, bias=b),
        "hipBLASLt rows @ wt + b": lambda: torch.addmm(b, rows, wt),
        "blocked_to_rows": lambda: ops.blocked_to_rows(u),
        "gemm_tn(dy, z) with column sums": lambda: ops.gemm_tn(rows, x_ext[:n_loc], colsum=True),
    }
    res = {k: [] for k in cases}
    for _ in range(3):
        for k, fn in cases.items():
            res[k].append(timed(fn))
    print(f"rank 0 of {P}, workload {wl['name']}: n_local {n_loc}, halo {f.n_halo}, edges {csr.nnz}")
    for k, v in res.items():
        print(f"{k:75s} {min(v):7.3f} ms (rounds: {' '.join(f'{t:.3f}' for t in v)})")


if __name__ == "__main__":
    main()
