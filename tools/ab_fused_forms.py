#!/usr/bin/env python3
"""A/B of rgbx_spmm_linear_f32 (same signature since round 2) between the in-tree build and another build of the
library (e.g. round 2's final kernel sources built into tools/ab/r02/librgbx_hip.so), form by form, on one box:
the forms a 2-layer GCN epoch launches — plain, ce_stats (loss statistics only), z+stats (aggregate stored + BatchNorm
column sums), pre+z+ce_grad (BatchNorm on the aggregate + aggregate stored + loss gradient written), and the plain form
over the transposed CSR — ALL gathering the same static matrix, interleaved A / B / plain-SpMM yardstick, HIP events.
Usage: python tools/ab_fused_forms.py <path to build B> [S|L] [rounds]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import WORKLOADS, synth
from rgb_experiment_amd import _lib, ops
from rgb_experiment_amd.graph import get_graph

P, I64 = ctypes.c_void_p, ctypes.c_int64
SIG = [P, P, P, P, P, I64, P, P, I64, P, P, P, I64, P, I64, P, P, P, P, P, ctypes.c_size_t, P, I64, I64, I64, P, P]


def bind(path):
    lib = ctypes.CDLL(path)
    lib.rgbx_spmm_linear_f32.restype = ctypes.c_int
    lib.rgbx_spmm_linear_f32.argtypes = SIG
    lib.rgbx_spmm_linear_stats_workspace_bytes.restype = ctypes.c_int
    lib.rgbx_spmm_linear_stats_workspace_bytes.argtypes = [I64, I64, ctypes.POINTER(ctypes.c_size_t)]
    lib.rgbx_last_error_string.restype = ctypes.c_char_p
    return lib


def main():
    path_b = sys.argv[1]
    wl = WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "L"]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    N, E, d = wl["N"], wl["E"], wl["d"]
    dev = torch.device("cuda:0")
    ei, x, y = synth(N, E, d)
    ei, x, y = ei.to(dev), x.to(dev), y.to(dev)
    g = get_graph(ei, N, 1)
    _ = g.bwd, g.w, g.w_t
    torch.manual_seed(0)
    wt = (torch.randn(d, d, device=dev) / d ** 0.5).contiguous()
    bias = torch.randn(d, device=dev)
    out, z = torch.empty(N, d, device=dev), torch.empty(N, d, device=dev)
    scale, shift = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
    rowsum = torch.rand(N, device=dev)
    mask = (torch.arange(N, device=dev) % 5 < 3).contiguous()
    gscale = torch.tensor([1.0 / N], device=dev)
    stats = torch.empty(3, dtype=torch.float64, device=dev)
    scratch = torch.empty(3 * ((N + 31) // 32 + 64), dtype=torch.float64, device=dev)
    colsums = torch.empty(2, d, dtype=torch.float64, device=dev)
    libs = {"A": bind(_lib.LIB_PATH), "B": bind(path_b)}
    nbytes = ctypes.c_size_t(0)
    libs["A"].rgbx_spmm_linear_stats_workspace_bytes(N, d, ctypes.byref(nbytes))
    nb = ctypes.c_size_t(0)
    libs["B"].rgbx_spmm_linear_stats_workspace_bytes(N, d, ctypes.byref(nb))
    ws = torch.empty(max(nbytes.value, nb.value, 8), dtype=torch.uint8, device=dev)
    ce_stats = _lib.CeEpilogue(y.data_ptr(), mask.data_ptr(), 0, stats.data_ptr(), scratch.data_ptr())
    ce_grad = _lib.CeEpilogue(y.data_ptr(), mask.data_ptr(), gscale.data_ptr(), stats.data_ptr(), scratch.data_ptr())
    stream = _lib.stream_ptr()

    def call(lib, csr, w, o=None, zz=None, pre=False, cs=False, ce=None):
        rc = lib.rgbx_spmm_linear_f32(csr.rowptr.data_ptr(), csr.col.data_ptr(), w.data_ptr(), None, x.data_ptr(), d,
                                      wt.data_ptr(), None, 0, None, bias.data_ptr(), None if o is None else o.data_ptr(), d,
                                      None if zz is None else zz.data_ptr(), d,
                                      scale.data_ptr() if pre else None, shift.data_ptr() if pre else None,
                                      rowsum.data_ptr() if pre else None, colsums.data_ptr() if cs else None,
                                      ws.data_ptr() if cs else None, ws.numel() if cs else 0,
                                      None if ce is None else ctypes.byref(ce), N, d, d, None, stream)
        if rc:
            raise RuntimeError(lib.rgbx_last_error_string().decode())

    forms = {
        "plain": lambda lib: call(lib, g.fwd, g.w, o=out),
        "ce_stats": lambda lib: call(lib, g.fwd, g.w, ce=ce_stats),
        "z+stats": lambda lib: call(lib, g.fwd, g.w, o=out, zz=z, cs=True),
        "pre+z+ce_grad": lambda lib: call(lib, g.fwd, g.w, o=out, zz=z, pre=True, ce=ce_grad),
        "plain, transposed CSR": lambda lib: call(lib, g.bwd, g.w_t, o=out),
    }

    def timed(fn, reps=5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    yard = lambda: ops.spmm_raw(g.fwd, g.w, None, x, out=out)
    res = {k: {"A": [], "B": [], "Y": []} for k in forms}
    for fn in forms.values():
        for lib in libs.values():
            fn(lib)
    yard()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in forms.items():
            res[k]["Y"].append(timed(yard))
            for tag, lib in libs.items():
                res[k][tag].append(timed(lambda: fn(lib)))
    med = lambda v: sorted(v)[len(v) // 2]
    table = {}
    for k, r in res.items():
        a, b, yv = med(r["A"]), med(r["B"]), med(r["Y"])
        table[k] = {"A_ms": a, "B_ms": b, "yardstick_ms": yv, "A_over_B": a / b, "A_over_yardstick": a / yv,
                    "B_over_yardstick": b / yv}
        print(f"{k:24s} A {a:7.3f} ms  B {b:7.3f} ms  A/B {a / b:6.3f}   plain SpMM {yv:7.3f} ms  A/Y {a / yv:6.3f}  "
              f"B/Y {b / yv:6.3f}", flush=True)
    print(json.dumps({"A": "in-tree build", "B": path_b, "workload": wl["name"], "rounds": rounds, "forms": table}))


if __name__ == "__main__":
    main()
