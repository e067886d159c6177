#!/usr/bin/env python3
"""Microbenchmark of the DENSE forms of rgbx_fused_layer_f32 (dense_stream_kernel / spmm_linear_kernel<DENSE>) and of
rgbx_gemm_tn_f32 against their bounds, at the row counts a partitioned rank (250 k) and a single GPU (2 M) run them:
  rows in -> rows out (+ bias)                       out = x W^T + b
  blocked in -> rows out, pre-affine + z             the training return stage without the loss
  rows in -> blocked out                             q = dy W
  rows in, root term                                 out = x W^T + x_r Wr^T + b
  loss statistics only / loss gradient               the last layer's return stage
  hipBLASLt (torch.addmm) as the yardstick
Floors printed per row: HBM = bytes / 8 TB/s, MFMA = 2 N K Nout / 157.3 TFLOP/s (fp32 MFMA peak).
Usage: python tools/dense_bench.py [other build of the library to A/B against] [rows ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rgb_experiment_amd import _lib, ops


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    args = sys.argv[1:]
    path_b = args.pop(0) if args and not args[0].isdigit() else None
    sizes = [int(a) for a in args] or [250_000, 2_000_000]
    dev = torch.device("cuda:0")
    libs = {"A": _lib.load()}
    if path_b:
        libs["B"] = _lib.bind(path_b, strict=False)
    out_rows = []
    for n in sizes:
        for d in (128, 64):
            torch.manual_seed(0)
            rows = torch.randn(n, d, device=dev)
            rows2 = torch.randn(n, d, device=dev)
            C = 4
            u = torch.randn((C, n, d // C), device=dev)
            blk = torch.empty((C, n, d // C), device=dev)
            wt = (torch.randn(d, d, device=dev) / d ** 0.5).contiguous()
            wtr = (torch.randn(d, d, device=dev) / d ** 0.5).contiguous()
            b = torch.randn(d, device=dev)
            s_, t_ = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
            rowsum = torch.rand(n, device=dev)
            y = torch.randint(0, d, (n,), device=dev)
            mask = (torch.arange(n, device=dev) % 5) < 3
            scale = ops.mask_scale(y, mask, d)
            o = torch.empty(n, d, device=dev)
            z = torch.empty(n, d, device=dev)
            mat = 4 * n * d
            cases = {
                "rows -> rows (+ bias)": (lambda: ops.fused_layer(rows, wt, bias=b, out=o), 2 * mat),
                "blocked -> rows, pre-affine, z stored": (lambda: ops.fused_layer(u, wt, bias=b, pre=(s_, t_, rowsum), out=o, z=z), 3 * mat),
                "rows -> blocked only": (lambda: ops.fused_layer(rows, wt, want_out=False, out_blocked=blk), 2 * mat),
                "rows -> rows, root term": (lambda: ops.fused_layer(rows, wt, bias=b, x_root=rows2, wt_root=wtr, out=o), 3 * mat),
                "rows -> rows + column sums": (lambda: ops.fused_layer(rows, wt, bias=b, out=o, want_colsums=True), 2 * mat),
                "loss statistics only": (lambda: ops.fused_layer(rows, wt, bias=b, ce=(y, mask, None)), mat),
                "loss gradient": (lambda: ops.fused_layer(rows, wt, bias=b, ce=(y, mask, scale), out=o), 2 * mat),
                "gemm_tn (dW = dy^T z) + column sums": (lambda: ops.gemm_tn(rows, rows2, colsum=True), 2 * mat),
                "hipBLASLt addmm": (lambda: torch.addmm(b, rows, wt, out=o), 2 * mat),
            }
            flops = 2.0 * n * d * d
            res = {k: {t: [] for t in libs} for k in cases}
            for _ in range(3):
                for k, (fn, _) in cases.items():
                    for tag, lib in libs.items():
                        _lib.use(lib)
                        res[k][tag].append(timed(fn, 10))
            _lib.use(libs["A"])
            for k, (fn, nbytes) in cases.items():
                fl = flops * (2 if "root" in k else 1)
                a = min(res[k]["A"])
                row = {"rows": n, "width": d, "form": k, "A_ms": a, "hbm_floor_ms": nbytes / 8e12 * 1e3,
                       "mfma_floor_ms": fl / 157.3e12 * 1e3, "tflops": fl / a / 1e9, "gbs": nbytes / a / 1e6}
                txt = f"n={n:8d} d={d:3d} {k:40s} A {a:7.3f} ms  {row['tflops']:6.1f} TF ({row['tflops'] / 157.3:4.2f} of MFMA)  " \
                      f"{row['gbs']:7.0f} GB/s ({row['gbs'] / 8000:4.2f} of HBM)"
                if "B" in libs:
                    row["B_ms"] = min(res[k]["B"])
                    txt += f"   B {row['B_ms']:7.3f} ms  A/B {a / row['B_ms']:5.3f}"
                print(txt, flush=True)
                out_rows.append(row)
            del rows, rows2, u, blk, o, z
            torch.cuda.empty_cache()
    print(json.dumps(out_rows))


if __name__ == "__main__":
    main()
