#!/usr/bin/env python3
"""bench.py — the reference's hot loop (itexperiments.py:417-473) on synthetic graphs, MI355X.

One STEP = one epoch of the reference loop body for a 2-layer GCN at d = 128: 1 train forward +
backward + Adam step, then 2 eval forwards (val, test) = 8 CSR SpMM launches (6 forward, 2
transposed) + 9 dense GEMMs + BatchNorm / log-softmax passes. Inputs are resident in HBM before the
timed region.

Metric (BASELINE.json): "aggregated edges/sec + training epochs/sec, full-graph GCN d=128".
  value            = edges aggregated per second over the WHOLE step = 8 * E' * steps / wall time
  epochs_per_s     = steps / wall time
  spmm_edges_per_s = E' / mean SpMM kernel time (HIP events on the launch stream)
  roofline         = algorithmic bytes of one SpMM / mean SpMM kernel time vs 8 TB/s HBM

Launch: `python bench.py [--gpus 1]`, or for N > 1 one rank per GPU under torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE from the environment); the graph is then 1-D node-partitioned over the
ranks with an RCCL all-to-all halo exchange per propagate (strong scaling: total work is fixed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

WORKLOADS = {
    # BASELINE.json configs[1] / the north-star target size (S and L)
    "S": dict(N=200_000, E=4_000_000, d=128, name="synthetic |V|=200k |E|=4M d=128, 2-layer GCN"),
    "L": dict(N=2_000_000, E=60_000_000, d=128, name="synthetic |V|=2M |E|=60M d=128, 2-layer GCN"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth(N, E, d):
    """directed iid-uniform endpoints (self-loops / duplicates left in), N(0,1) features,
    uniform labels over d classes; fixed seeds."""
    ei = torch.randint(0, N, (2, E), generator=torch.Generator().manual_seed(1234567), dtype=torch.int64)
    x = torch.randn(N, d, generator=torch.Generator().manual_seed(1234568))
    y = torch.randint(0, d, (N,), generator=torch.Generator().manual_seed(1234569))
    return ei, x, y


def split_masks(N):
    """6-2-2 split. The reference's get_whole_mask shuffles a Python list of N ints (seconds at 2M
    nodes, outside the timed region either way); a seeded permutation gives the same row counts."""
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(123456789))
    a, b = int(0.6 * N), int(0.6 * N) + int(0.2 * N)
    masks = []
    for part in (perm[:a], perm[a:b], perm[b:]):
        m = torch.zeros(N, dtype=torch.bool)
        m[part] = True
        masks.append(m)
    return masks


def pmc_traffic(workload, world):
    """HBM-side bytes per SpMM launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, gfx950 correction; profiles/pmc_traffic_<workload>.json). PMC counters cannot be read
    from inside this process, so this is the last profiled value for the same kernel and workload, or None."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["traffic_bytes_per_launch"]


def spmm_alg_bytes(n_rows, nnz, d):
    """gathered rows + col + weight per edge, output row + rowptr per node."""
    return nnz * (4 * d + 8) + n_rows * 4 * d + 4 * (n_rows + 1)


def cpu_baseline(ei, x, N, budget_s=20.0):
    """The oracle's PyG-style dataflow (index_select -> multiply -> index_add_) on the host cores, on a
    bounded sample: the first M rewritten edges of ONE GCN propagate at d = 128."""
    from oracle import ref_cpu as O
    torch.set_num_threads(os.cpu_count() or 1)
    rei, w = O.gcn_norm(ei, None, N)
    M = min(rei.size(1), 8_000_000)
    sub, wsub = rei[:, :M].contiguous(), w[:M].contiguous()
    O.propagate(sub[:, :100_000], x, N, wsub[:100_000], "add")  # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 10):
        t0 = time.perf_counter()
        O.propagate(sub, x, N, wsub, "add")
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": M / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.propagate (index_select*w -> index_add_) over the first {M} of the {rei.size(1)} "
                      f"rewritten edges of one GCN propagate, d=128, median of {len(times)} runs",
            "seconds_per_run": med}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="L")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("bench.py: --gpus N > 1 must be launched with `python -m torch.distributed.run "
                 "--nproc-per-node N ...` (one rank per GPU)")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RGBX_DIST_BACKEND=gloo rehearses the N>1 code path with several ranks on ONE GPU (host-staged
        # collectives); the real runs use RCCL ("nccl").
        backend = os.environ.get("RGBX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from rgb_experiment_amd import ops
    from rgb_experiment_amd.models import GCN

    wl = WORKLOADS[args.workload]
    N, E, d = wl["N"], wl["E"], wl["d"]
    ei, x, y = synth(N, E, d)
    train_mask, val_mask, test_mask = split_masks(N)

    torch.manual_seed(14530529)  # the reference's reappear_seed (itexperiments.py:57)
    model = GCN(num_layers=2, hidden_unit=d, input_dim=d, output_dim=d, dropout_rate=0.5)

    if world > 1:
        from rgb_experiment_amd.dist import DistGCNRunner
        from rgb_experiment_amd.graph import LOOPS_ADD_REMAINING
        runner = DistGCNRunner(model, ei, x, y, (train_mask, val_mask, test_mask), rank, world, dev, lr=0.01)
        plan = runner.plan(LOOPS_ADD_REMAINING, "gcn")  # partition + per-rank CSRs built here, once
        nnz_total = plan.nnz_total
        step = runner.epoch
        n_loc = plan.n_local
        nnz_loc, nnz_rem = int(plan.fwd.loc_agg.numel()), int(plan.fwd.rem_agg.numel())
        # one propagate on a rank = local-source SpMM + remote-source SpMM accumulating into the same rows
        alg = spmm_alg_bytes(n_loc, nnz_loc, d) + spmm_alg_bytes(n_loc, nnz_rem, d) + n_loc * 4 * d
        halo_mb = plan.fwd.n_halo * d * 4 / 1e6
    else:
        from rgb_experiment_amd.graph import get_graph, LOOPS_ADD_REMAINING
        model.to(dev)
        ei_d, x_d, y_d = ei.to(dev), x.to(dev), y.to(dev)
        tm, vm, sm = train_mask.to(dev), val_mask.to(dev), test_mask.to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        graph = get_graph(ei_d, N, LOOPS_ADD_REMAINING)
        _ = graph.w, graph.w_t  # graph preparation happens once per edge_index, outside the loop
        nnz_total = graph.fwd.nnz
        alg = spmm_alg_bytes(N, nnz_total, d)
        halo_mb = 0.0

        def evaluate(mask):
            model.eval()
            with torch.no_grad():
                out = model(x_d, ei_d)["out"]
            s = ops.masked_nll_accuracy(out, y_d, mask).tolist()  # NLLLoss on out[mask] + arg-max accuracy
            return s[0] / s[1], s[2] / s[1]

        def step():
            model.train()
            opt.zero_grad()
            out = model(x_d, ei_d)["out"]
            loss = ops.masked_nll_loss(out, y_d, tm)
            train_loss = loss.item()
            loss.backward()
            opt.step()
            return (train_loss,) + evaluate(vm) + evaluate(sm)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = []
    ops.set_event_sink(events)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.set_event_sink(None)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    n_prop = 8  # 6 forward + 2 transposed propagates per epoch (2-layer GCN)
    kinds = ("gcn_fwd", "gcn_bwd") if world == 1 else ("dist_fwd_local", "dist_fwd_remote", "dist_bwd_local",
                                                       "dist_bwd_remote")
    spmm_total_ms = sum(s.elapsed_time(e) for kind, s, e in events if kind in kinds)
    spmm_avg_s = spmm_total_ms * 1e-3 / (n_prop * args.steps)  # kernel time per propagate (on this rank)
    achieved = alg / spmm_avg_s / 1e9
    by_kind = {}
    for kind, s, e in events:
        by_kind.setdefault(kind, []).append(s.elapsed_time(e))
    by_kind = {k: {"n": len(v), "avg_ms": sum(v) / len(v)} for k, v in sorted(by_kind.items())}

    result = {
        "metric": "aggregated edges/sec (full-graph GCN d=128, reference epoch = train fwd+bwd+Adam + 2 eval fwd)",
        "value": n_prop * nnz_total * args.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["name"], "nodes": N, "edges_in": E, "edges_aggregated_per_propagate": nnz_total,
                   "width": d, "propagates_per_step": n_prop,
                   "parallelism": "single GPU" if world == 1 else f"1-D node partition x{world}, RCCL all-to-all halo"},
        "epochs_per_s": args.steps / elapsed,
        "spmm_edges_per_s": nnz_total / spmm_avg_s if world == 1 else None,
        "spmm_ms": spmm_avg_s * 1e3,
        "halo_mb_per_rank_per_propagate": halo_mb,
        "kernel_ms_by_kind": by_kind,
        "final_losses": {"train": last[0], "val": last[1], "test": last[3]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, world),
                     "kernel": "spmm_csr_kernel<32,4,true>", "algorithmic_bytes_per_launch": alg,
                     "note": "rank 0's share (local + remote SpMM)" if world > 1 else "whole graph"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(ei, x, N)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
