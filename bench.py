#!/usr/bin/env python3
"""bench.py — the reference's hot loop (itexperiments.py:417-473) on synthetic graphs, MI355X.

One STEP = one epoch of the reference loop body for a 2-layer GCN at d = 128: 1 train forward +
backward + Adam step, then 2 eval forwards (val, test) = 7 aggregations over all E' edges (6 forward, 1
transposed: the input layer aggregates first, so its weight gradient needs no pass through A_hat^T), each
fused with its layer's dense transform (fp32 MFMA) where the shapes allow, + the weight-gradient GEMMs,
BatchNorm and the loss. `value` counts the edges actually aggregated (7 * E' per step), not the 8 propagates
of the transform-first formulation. Nothing of the epoch is skipped or cached across steps; two things are
computed in a different FORM than the reference writes them, with the same values: the eval-mode BatchNorm is
folded into the preceding layer's weights, and loss / accuracy / loss gradient are taken from the logits
(cross-entropy = NLLLoss o log_softmax) so the log-probabilities are never written out.
The five numbers the reference reads with .item() inside the loop body are read once, at the end of the step.
Inputs are resident in HBM before the timed region.

Metric (BASELINE.json): "aggregated edges/sec + training epochs/sec, full-graph GCN d=128".
  value            = edges aggregated per second over the WHOLE step = 7 * E' * steps / wall time
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
# dmabuf IPC for RCCL between the ranks of one node: must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def gat_alg_bytes(n_rows, nnz, d, H=8):
    """Forward: col + a_src[H] + feature row per edge, out + max + 1/sum per node. Backward: the source
    pass (col + 16H-byte record + gout row per edge, ds store), the width-H segment sum, the streaming prep.
    Returned: the mean over the 6 forward and 2 backward propagates of an epoch."""
    fwd = nnz * (4 + 4 * H + 4 * d) + n_rows * (8 * d + 12 * H) + 4 * (n_rows + 1)
    bwd = nnz * (4 + 16 * H + 4 * d) + nnz * 4 * H + nnz * (4 + 4 * H) + n_rows * (16 * d + 24 * H)
    return (6 * fwd + 2 * bwd) / 8


def spmm_alg_bytes(n_rows, nnz, d):
    """gathered rows + col + weight per edge, output row + rowptr per node."""
    return nnz * (4 * d + 8) + n_rows * 4 * d + 4 * (n_rows + 1)


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota if there is one (a one-GPU box grants
    a share of the host, not all of os.cpu_count()), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(ei, x, N, budget_s=20.0, hip_propagate=None):
    """The oracle timed on the host cores, one GCN propagate at d = 128 (rank 0, N = 1 only).
    Primary: the C/OpenMP restatement (oracle/propagate_ref.c, per-target CSR sums over all host threads)
    on the WHOLE rewritten edge list — the strongest plain CPU form of the same arithmetic.
    Also reported: the PyG-style dataflow of the Python oracle (index_select -> multiply -> index_add_,
    which materialises [E, d]) on a bounded 8 M-edge sample, which is what the reference's CPU path does.
    `hip_propagate` (the HIP kernel's result of the same propagate, on the host) is checked against the C result:
    the full-size parity of this very run rides along on the line."""
    from oracle import ref_cpu as O
    cores = host_cores()
    torch.set_num_threads(cores)
    rei, w = O.gcn_norm(ei, None, N)
    M = min(rei.size(1), 8_000_000)
    sub, wsub = rei[:, :M].contiguous(), w[:M].contiguous()
    O.propagate(sub[:, :100_000], x, N, wsub[:100_000], "add")  # warm-up
    times = []
    t_end = time.perf_counter() + budget_s / 2
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 6):
        t0 = time.perf_counter()
        O.propagate(sub, x, N, wsub, "add")
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    dataflow = {"value": M / med, "unit": "edges/s", "cores": torch.get_num_threads(), "seconds_per_run": med,
                "what": f"oracle.propagate (index_select*w -> index_add_, the PyG dataflow) over the first {M} of "
                        f"the {rei.size(1)} rewritten edges, median of {len(times)} runs"}
    try:
        rowptr, col, perm = O.csr_from_edges(rei[1], rei[0], torch.arange(rei.size(1)), N)
        ws = w[perm.long()].contiguous()
        threads = min(O.c_threads(), cores)
        cpu_out = O.propagate_c_csr(rowptr, col, ws, x, "add", threads)  # warm-up (page faults, thread pool)
        parity = None
        if hip_propagate is not None:
            parity = {"max_abs_diff_hip_vs_cpu": (hip_propagate - cpu_out).abs().max().item(),
                      "max_abs_value": cpu_out.abs().max().item(), "rows": N, "width": x.size(1)}
        del cpu_out
        t2 = []
        t_end = time.perf_counter() + budget_s / 2
        while len(t2) < 3 or (time.perf_counter() < t_end and len(t2) < 20):
            t0 = time.perf_counter()
            O.propagate_c_csr(rowptr, col, ws, x, "add", threads)
            t2.append(time.perf_counter() - t0)
        t2.sort()
        m2 = t2[len(t2) // 2]
        return {"value": rei.size(1) / m2, "unit": "edges/s", "cores": threads, "kind": "port",
                "sample": f"oracle/propagate_ref.c oracle_propagate_csr_f32 (C + OpenMP, {threads} threads) over all "
                          f"{rei.size(1)} rewritten edges of one GCN propagate, d=128, median of {len(t2)} runs",
                "seconds_per_run": m2, "pyg_dataflow_variant": dataflow, "parity_at_full_size": parity}
    except Exception as exc:  # C restatement not built: fall back to the Python oracle's number
        dataflow.update({"kind": "port", "sample": dataflow.pop("what"), "c_restatement_error": repr(exc)})
        return dataflow


MODELS = {
    # name: (constructor kwargs, propagates per epoch = 3 forwards + 1 backward, loops_mode, weighting kind)
    # gcn: 2 + 2 + 2 forward SpMMs and ONE transposed SpMM (layer 2); layer 1's weight gradient is taken
    # against A_hat x, so nothing flows back through A_hat^T there (nn/conv.py GCNConv)
    "gcn": (dict(num_layers=2, hidden_unit=128, dropout_rate=0.5), 7, 1, "gcn"),
    "graphsage": (dict(num_layers=2, hidden_unit=128, dropout_rate=0.5), 7, 2, "mean"),
    "graphsage2": (dict(num_layers=2, hidden_unit=128, dropout_rate=0.5), 7, 0, "mean"),
    "gat": (dict(num_layers=2, hidden_unit=16, heads=8, dropout_rate=0.5), 8, 2, "gat"),
    "appnpstack": (dict(hidden_unit=64, K=10, alpha=0.1, dropout_rate=0.5), 40, 1, "gcn"),
}
AGG_KINDS = ("gcn_fwd", "gcn_bwd", "mean_fwd", "mean_bwd", "gcn_linear_fwd", "mean_linear_fwd", "gcn_linear_bwd",
             "mean_linear_bwd", "appnp_fwd",
             "appnp_bwd", "gat_fwd", "gat_bwd_prep", "gat_bwd_src", "gat_bwd_segsum", "dist_fwd_local", "dist_fwd_remote",
             "dist_bwd_local", "dist_bwd_remote", "dist_fwd_resident",
             "dist_fwd_colshard", "dist_bwd_colshard", "dist_fwd_appnp_colshard", "dist_bwd_appnp_colshard")


def build_single_gpu(model, ei, x, y, masks, dev, loops_mode, kind, N, d):
    """One-GPU step function = the reference loop body (itexperiments.py:417-473): train forward +
    backward + Adam, then the val and test eval forwards. Returns (step, E' per propagate, algorithmic
    bytes of one aggregation launch)."""
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import get_graph
    model.to(dev)
    ei_d, x_d, y_d = ei.to(dev), x.to(dev), y.to(dev)
    tm, vm, sm = (m.to(dev) for m in masks)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    graph = get_graph(ei_d, N, loops_mode)  # graph preparation happens once per edge_index, outside the loop
    _ = graph.bwd
    if kind == "gcn":
        _ = graph.w, graph.w_t
    nnz_total = graph.fwd.nnz
    if kind == "gat":  # SURVEY §8d per-launch bytes, averaged over the 6 forward + 2 backward propagates
        alg = gat_alg_bytes(N, nnz_total, d)
    else:
        alg = spmm_alg_bytes(N, nnz_total, d)

    def evaluate(mask):
        model.eval()
        with torch.no_grad():
            logits = model(x_d, ei_d)["emb"]
        # device [nll sum, count, correct] = NLLLoss on log_softmax(logits)[mask] + arg-max accuracy, from the logits
        return ops.masked_ce_accuracy(logits, y_d, mask)

    def step():
        """The numbers the reference reads with .item() at three points of the loop body (itexperiments.py:437,
        467, 472) are only used after the epoch (early stopping, curves): they stay on the device and come back in
        ONE copy at the end, so the GPU queue does not drain three times per epoch."""
        model.train()
        opt.zero_grad()
        loss = ops.masked_ce_loss(model(x_d, ei_d)["emb"], y_d, tm)  # = NLLLoss(log_softmax(.)[mask], y[mask])
        loss.backward()
        opt.step()
        val, tst = evaluate(vm), evaluate(sm)
        s = torch.cat([loss.detach().double().reshape(1), val, tst]).tolist()  # the one host sync of the epoch
        return s[0], s[1] / s[2], s[3] / s[2], s[4] / s[5], s[6] / s[5]

    def graphed():
        """The same epoch captured once as a hipGraph and replayed (rgb_experiment_amd.epoch_graph): needs a
        fresh capturable Adam; returns a run() callable."""
        from rgb_experiment_amd.epoch_graph import GraphedEpoch
        gopt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
        return GraphedEpoch(model, gopt, {"x": x_d, "edge_index": ei_d}, y_d, (tm, vm, sm)).capture().run

    step.graphed = graphed
    return step, nnz_total, alg


def time_graphed(step, steps, warmup):
    """ms per epoch of the hipGraph replay of the same epoch (secondary number; `value` stays on the eager
    loop, whose launches carry the per-kernel HIP events the roofline needs)."""
    try:
        run = step.graphed()
        for _ in range(warmup):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"ms_per_step": dt / steps * 1e3, "epochs_per_s": steps / dt}
    except Exception as exc:
        return {"error": repr(exc)}


def secondary_config(dev, steps, warmup):
    """BASELINE.json configs[1] (|V|=200k, |E|=4M, 2-layer GCN, one GPU) measured in the same process, so
    that both readings of "the configuration the metric is quoted on" are on the line: the headline value
    is the north-star target size L, this block is S. Note X (102 MB) sits in the 256 MiB Infinity Cache at
    S, so its roofline fraction is cache-served (SURVEY §8d caveat)."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import clear_cache
    wl = WORKLOADS["S"]
    N, E, d = wl["N"], wl["E"], wl["d"]
    ei, x, y = synth(N, E, d)
    torch.manual_seed(14530529)
    model = M.GCN(input_dim=d, output_dim=d, **MODELS["gcn"][0])
    step, nnz, alg = build_single_gpu(model, ei, x, y, split_masks(N), dev, 1, "gcn", N, d)
    for _ in range(warmup):
        step()
    events = []
    ops.set_event_sink(events)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_event_sink(None)
    n_prop = MODELS["gcn"][1]
    kinds = ("gcn_fwd", "gcn_bwd", "gcn_linear_fwd", "gcn_linear_bwd")
    spmm_s = sum(s.elapsed_time(e) for k, s, e in events if k in kinds) * 1e-3 / (n_prop * steps)
    replay = time_graphed(step, steps, warmup)
    clear_cache()
    return {"workload": wl["name"], "value": n_prop * nnz * steps / elapsed, "unit": "edges/s",
            "ms_per_step": elapsed / steps * 1e3, "epochs_per_s": steps / elapsed, "spmm_ms": spmm_s * 1e3,
            "hip_graph_replay": replay,
            "roofline_frac_algorithmic": alg / spmm_s / 1e9 / HBM_PEAK_GBS,
            "note": "X fits the Infinity Cache at this size: the fraction is cache-served, not HBM"}


def launch_ranks(n):
    """`python bench.py --gpus N` as typed: start N ranks (one per GPU) with torch.distributed.run as CHILD
    processes and return their exit status. Rank 0 of the children prints the JSON line on the inherited stdout.
    Called before anything in this process has touched the GPU; this process never execs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="L")
    ap.add_argument("--model", choices=sorted(MODELS), default="gcn",
                    help="gcn is the BASELINE.json headline; the others are its configs 3-5")
    ap.add_argument("--exchange", choices=["auto", "halo", "reshard"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--primary-only", action="store_true",
                    help="skip the secondary legs (hipGraph replay, configs[1] block): clean rocprofv3 runs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as `python bench.py --gpus N`: this process has made no GPU call yet; it only starts the N ranks
        # as children (one per GPU) and leaves with their status — it never replaces itself
        sys.exit(launch_ranks(args.gpus))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("RGBX_DIST_BACKEND", "nccl")
    test_backend = None
    if backend == "gloo" and not torch.cuda.is_available():
        # CPU rehearsal of the N > 1 host logic (tests/test_bench_cli.py): gloo collectives and an aggregator the TEST
        # injects; the product has no CPU aggregation path of its own, so without one this is an error
        spec = os.environ.get("RGBX_TEST_AGGREGATOR")
        if world == 1 or not spec:
            sys.exit("bench.py: no MI355X visible; the only CPU mode is the gloo rehearsal of --gpus N > 1 with "
                     "RGBX_TEST_AGGREGATOR=module:Class set by a test")
        mod, cls_name = spec.split(":")
        test_backend = getattr(__import__(mod, fromlist=[cls_name]), cls_name)()
        dev = torch.device("cpu")
    else:
        local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        # RGBX_DIST_BACKEND=gloo rehearses the N>1 code path with several ranks on ONE GPU (host-staged
        # collectives) or on the CPU (above); the real runs use RCCL ("nccl").
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from rgb_experiment_amd import models as M
    from rgb_experiment_amd import ops

    wl = WORKLOADS[args.workload]
    N, E, d = wl["N"], wl["E"], wl["d"]
    ei, x, y = synth(N, E, d)
    train_mask, val_mask, test_mask = split_masks(N)
    kwargs, n_prop, loops_mode, kind = MODELS[args.model]
    cls = {"gcn": M.GCN, "graphsage": M.GraphSAGE, "graphsage2": M.GraphSAGE2, "gat": M.GAT,
           "appnpstack": M.APPNPStack}[args.model]

    torch.manual_seed(14530529)  # the reference's reappear_seed (itexperiments.py:57)
    model = cls(input_dim=d, output_dim=d, **kwargs)
    wl_name = wl["name"].replace("GCN", {"gcn": "GCN", "graphsage": "GraphSAGE", "graphsage2": "GraphSAGE2",
                                         "gat": "GAT 8 heads", "appnpstack": "APPNP K=10"}[args.model])

    comm_mb, scheme, alg_by_kind = 0.0, "single GPU", None
    if world > 1:
        from rgb_experiment_amd.dist import DistRunner
        runner = DistRunner(model, ei, x, y, (train_mask, val_mask, test_mask), rank, world, dev, lr=0.01,
                            exchange=args.exchange)
        dgraph = runner.graphs[loops_mode]
        step = runner.epoch
        n_loc = runner.hi - runner.lo
        scheme = dgraph.scheme(d) if kind != "gat" else "halo"
        if kind == "gat":  # halo rows appended to the local rows, one rectangular CSR
            model.eval()
            with torch.no_grad():
                model(runner.x, runner.token)  # builds the plan + rectangular CSRs once, outside the timing
            plan = dgraph._kinds["gat"]["plan"]
            nnz_total = plan.nnz_total
            alg = gat_alg_bytes(n_loc, plan.nnz_local, d)
            comm_mb = plan.fwd.n_halo * d * 4 / 1e6
        else:
            # algorithmic bytes per launch of every aggregation kernel this rank runs (SURVEY 8d formula on the
            # rows / edges / width that launch covers); the roofline line is total bytes / total kernel time
            plan = dgraph.plan(kind)
            nnz_total = plan.nnz_total
            nnz_loc, nnz_rem = int(plan.fwd.loc_agg.numel()), int(plan.fwd.rem_agg.numel())
            b_loc = spmm_alg_bytes(n_loc, nnz_loc, d)
            b_rem = spmm_alg_bytes(n_loc, nnz_rem, d) + n_loc * 4 * d
            b_col_full = spmm_alg_bytes(N, nnz_total, d // world)  # whole graph on this rank's d / P columns
            # the column-shard SpMM runs as reshard_chunks * P launches (one per piece of every peer's row block,
            # so that the transpose back overlaps it): bytes per launch accordingly
            pieces = max(int(dgraph._chunks()), 1)
            b_col = b_col_full / (pieces * world if pieces > 1 else 1)
            b_res = spmm_alg_bytes(n_loc, nnz_loc + nnz_rem, d)  # first conv: resident [local; halo] features
            alg_by_kind = {"dist_fwd_resident": b_res, "gcn_linear_fwd": b_res, "mean_linear_fwd": b_res,
                           "dist_fwd_local": b_loc, "dist_bwd_local": b_loc, "dist_fwd_remote": b_rem,
                           "dist_bwd_remote": b_rem, "dist_fwd_colshard": b_col, "dist_bwd_colshard": b_col,
                           "dist_fwd_appnp_colshard": 10 * (b_col_full + N * 4 * (d // world)),
                           "dist_bwd_appnp_colshard": 10 * (b_col_full + N * 4 * (d // world))}
            alg = b_col_full if scheme == "reshard" else b_loc + b_rem
            comm_mb = (2 * n_loc * d * 4 * (world - 1) / world if scheme == "reshard" else plan.fwd.n_halo * d * 4) / 1e6
    else:
        step, nnz_total, alg = build_single_gpu(model, ei, x, y, (train_mask, val_mask, test_mask), dev, loops_mode,
                                                kind, N, d)

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

    agg_total_ms = sum(s.elapsed_time(e) for k, s, e in events if k in AGG_KINDS)
    agg_avg_s = agg_total_ms * 1e-3 / (n_prop * args.steps)  # aggregation kernel time per propagate (this rank)
    achieved = alg / agg_avg_s / 1e9
    if alg_by_kind:  # partitioned run: launches of different shapes, so sum bytes over the launches actually made
        done = sum(alg_by_kind[k] for k, s, e in events if k in alg_by_kind)
        achieved = done / (agg_total_ms * 1e-3) / 1e9
    by_kind = {}
    for k, s, e in events:
        by_kind.setdefault(k, []).append(s.elapsed_time(e))
    by_kind = {k: {"n": len(v), "avg_ms": sum(v) / len(v)} for k, v in sorted(by_kind.items())}
    kernel = {"gat": "gat_fwd_kernel<4> / gat_bwd_src_kernel<4> (+ prep, segment sum)"}.get(
        args.model, "spmm_linear_kernel<32,*,128,1> (aggregate + fp32 MFMA transform; forward, and backward on the transposed CSR)"
        if scheme != "reshard" else f"spmm_csr_kernel at width {d // world}")

    result = {
        "metric": "aggregated edges/sec (full-graph GCN d=128, reference epoch = train fwd+bwd+Adam + 2 eval fwd)"
                  if args.model == "gcn" else f"aggregated edges/sec (full-graph {args.model} d=128, reference epoch)",
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
        "config": {"workload": wl_name, "nodes": N, "edges_in": E, "edges_aggregated_per_propagate": nnz_total,
                   "width": d, "propagates_per_step": n_prop,
                   "parallelism": "single GPU" if world == 1 else
                   f"1-D node partition x{world}, RCCL all-to-all ({scheme} exchange; boundary rows of the static "
                   "input features resident in HBM)"},
        "epochs_per_s": args.steps / elapsed,
        "spmm_edges_per_s": nnz_total / agg_avg_s if world == 1 else None,
        "spmm_ms": agg_avg_s * 1e3,
        "exchange_mb_per_rank_per_propagate": comm_mb,
        "kernel_ms_by_kind": by_kind,
        "final_losses": {"train": last[0], "val": last[1], "test": last[3]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(args.workload, world) if args.model == "gcn" else None,
                     "kernel": kernel, "algorithmic_bytes_per_launch": alg,
                     "note": "rank 0's share of one propagate" if world > 1 else "whole graph, one propagate"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from rgb_experiment_amd.graph import get_graph
        with torch.no_grad():  # one GCN propagate of the whole workload by the HIP kernel, for the parity field
            hip_out = ops.propagate_gcn(x.to(dev), get_graph(ei.to(dev), N, 1)).cpu()
        result["cpu_baseline"] = cpu_baseline(ei, x, N, hip_propagate=hip_out)
        del hip_out
    if world == 1 and not args.primary_only:
        result["hip_graph_replay"] = time_graphed(step, args.steps, args.warmup)
    if world == 1 and args.workload == "L" and args.model == "gcn" and not args.primary_only:
        del step, model
        from rgb_experiment_amd.graph import clear_cache
        clear_cache()
        torch.cuda.empty_cache()
        result["configs_1_same_run"] = secondary_config(dev, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
