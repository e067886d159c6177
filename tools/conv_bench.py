#!/usr/bin/env python3
"""Per-layer timing of every conv type on the benchmark graphs (HIP events), forward and backward, with the
algorithmic-byte roofline of SURVEY §8d. Usage: python tools/conv_bench.py [S|L]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import WORKLOADS, synth
from rgb_experiment_amd import ops
from rgb_experiment_amd.graph import LOOPS_ADD_REMAINING, LOOPS_REMOVE_ADD, get_graph


def timed(fn, reps=10):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "S"]
    N, E, d = wl["N"], wl["E"], wl["d"]
    dev = torch.device("cuda:0")
    ei, x, _ = synth(N, E, d)
    ei, x = ei.to(dev), x.to(dev)
    gy = torch.randn_like(x)
    rows = []

    g1 = get_graph(ei, N, LOOPS_ADD_REMAINING)
    nnz = g1.fwd.nnz
    b_spmm = nnz * (4 * d + 8) + N * 4 * d + 4 * (N + 1)
    rows.append(("gcn spmm fwd", timed(lambda: ops.spmm_raw(g1.fwd, g1.w, None, x)), b_spmm))
    rows.append(("gcn spmm bwd", timed(lambda: ops.spmm_raw(g1.bwd, g1.w_t, None, gy)), b_spmm))
    g2 = get_graph(ei, N, LOOPS_REMOVE_ADD)
    b_mean = nnz * (4 * d + 4) + N * 4 * d + 4 * (N + 1) + 4 * N
    rows.append(("sage mean fwd (mode 2)", timed(lambda: ops.spmm_raw(g2.fwd, None, g2.inv_deg, x)), b_mean))
    rows.append(("sage mean bwd", timed(lambda: ops.spmm_raw(g2.bwd, g2.w_mean_t, None, gy)), b_spmm))
    rows.append(("appnp K=10 fwd", timed(lambda: ops.appnp_raw(g1.fwd, g1.w, x, 10, 0.1), 3), 10 * (b_spmm + N * 4 * d)))

    for H, C in ((8, 16), (1, 128)):
        a_s = torch.randn(N, H, device=dev)
        a_d = torch.randn(N, H, device=dev)
        hx = x.clone().requires_grad_(True)
        out = ops.gat_aggregate(hx, a_s.requires_grad_(True), a_d.requires_grad_(True), g2, H, C, 0.2)
        b_fwd = nnz * (4 + 4 * H + 4 * H * C) + N * (4 * H * C + 4 * H) + N * (4 * H * C + 8 * H) + 4 * (N + 1)
        att = torch.randn(1, H, C, device=dev)
        rows.append((f"gat fwd H={H} C={C} (a_src gathered)",
                     timed(lambda: ops.gat_aggregate(x, a_s.detach(), a_d.detach(), g2, H, C, 0.2)), b_fwd))
        rows.append((f"gat fwd H={H} C={C} (a_src in-kernel)",
                     timed(lambda: ops.gat_aggregate(x, a_s.detach(), a_d.detach(), g2, H, C, 0.2, att_src=att)),
                     b_fwd - nnz * 4 * H))
        t_bwd = timed(lambda: torch.autograd.grad(out, (hx, a_s, a_d), gy, retain_graph=True))
        # prep (stream out, gout) + source pass (col, 16H-byte record, gout row per edge; ds store) + segment sum
        b_bwd = nnz * (4 + 16 * H + 4 * H * C) + nnz * 4 * H + nnz * (4 + 4 * H) + N * (16 * H * C + 24 * H)
        rows.append((f"gat bwd (prep+src+segsum) H={H} C={C}", t_bwd, b_bwd))
    print(f"workload {wl['name']}  E'={nnz}")
    for name, ms, b in rows:
        print(f"{name:34s} {ms:9.3f} ms   {b / 1e9:8.2f} GB alg -> {b / ms / 1e6:8.1f} GB/s = {b / ms / 1e6 / 8000:5.3f} of 8 TB/s"
              f"   {nnz / ms / 1e6:8.2f} G edges/s")


if __name__ == "__main__":
    main()
