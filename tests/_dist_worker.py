"""Worker processes for the world_size>1 tests (gloo, CPU). The distributed HOST logic under test is
the product's (plan, halo exchange, DistBatchNorm, gradient all-reduce, runner); the per-rank
aggregation arithmetic is injected from the oracle because there is no GPU here."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_cpu as O  # noqa: E402


class OracleAggregator:
    """Same interface as rgb_experiment_amd.dist.HipAggregator, computed by oracle.propagate."""

    def prepare(self, agg, gather, n_rows, w):
        return torch.stack([gather, agg]), n_rows, w

    def run(self, handle, x, y=None, kind=None):
        ei, n_rows, w = handle
        out = O.propagate(ei, x, n_rows, w, "add")
        return out if y is None else y.add_(out)

    def run_rows(self, handle, x, lo, hi, out, kind=None, accumulate=False):
        rows = self.run(handle, x)[lo:hi]
        out.add_(rows) if accumulate else out.copy_(rows)
        return True

    def gather(self, x, idx):
        return x[idx.long()]

    def scatter_add(self, src, idx, dst):
        return dst.index_add_(0, idx.long(), src)

    def appnp(self, handle, h, K, alpha, kind=None):
        z = h
        for _ in range(K):
            z = (1 - alpha) * self.run(handle, z) + alpha * h
        return z

    def prepare_rect(self, agg, gather, n_tgt, n_src):
        return torch.stack([gather, agg]), n_tgt

    def gat(self, rect, x_ext, att_src, att_dst, n_tgt, H, C, slope):
        """GATConv attention on a rectangular graph (targets = the first n_tgt rows), plain torch ops."""
        ei, _ = rect
        h = x_ext.view(-1, H, C)
        a_s = (h * att_src.view(1, H, C)).sum(-1)
        a_d = (h[:n_tgt] * att_dst.view(1, H, C)).sum(-1)
        src, dst = ei[0], ei[1]
        e = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], slope)
        alpha = O.segment_softmax(e, dst, n_tgt)
        out = torch.zeros((n_tgt, H, C), dtype=x_ext.dtype).index_add_(0, dst, h[src] * alpha.unsqueeze(-1))
        return out.reshape(n_tgt, H * C)


def _rows_of(blocked):
    """[B, n, cols] blocked -> [n, B * cols]."""
    return blocked.permute(1, 0, 2).reshape(blocked.size(1), -1)


def _store_blocked(dst, rows):
    B, n, c = dst.shape
    dst.copy_(rows.view(n, B, c).permute(1, 0, 2))


class TorchStackBackend:
    """What dist.stack.HipStackBackend computes (rgbx_fused_layer_f32 & co.), in plain torch on the CPU, with the
    aggregation from the oracle: the checker of GridStack's host logic (layouts, exchange views, manual backward)."""

    def __init__(self, agg):
        self.agg = agg

    def layer(self, x, wt, handle=None, rows=None, bias=None, x_root=None, wt_root=None, pre=None, want_out=True,
              out_blocked=None, want_z=False, want_colsums=False, ce=None, kind=None, out=None, z=None):
        out_arg, z_arg = out, z
        if x.dim() == 3:
            assert handle is None
            x = _rows_of(x)
        z = x if handle is None else self.agg.run(handle, x)
        if rows is not None:
            z = z[rows[0]:rows[1]]
        if pre is not None:
            scale, shift, rowsum = pre
            z = z * scale + shift * rowsum[:, None]
        out = z @ wt
        if bias is not None:
            out = out + bias
        if wt_root is not None:
            xr = _rows_of(x_root) if x_root.dim() == 3 else x_root
            if pre is not None:
                xr = xr * pre[0] + pre[1]
            out = out + xr @ wt_root
        extra = None
        if want_colsums:
            od = out.double()
            extra = torch.stack([od.sum(0), (od * od).sum(0)])
        if out_blocked is not None:
            _store_blocked(out_blocked, out)
        if ce is not None and isinstance(ce[1], (tuple, list)):  # two masks, statistics only: [2, 3]
            extra = torch.stack([self.ce_stats(out, ce[0], m) for m in ce[1]])
            out = None
        elif ce is not None:
            y, mask, grad_scale = ce
            sel = (y >= 0) & (y < out.size(1))
            if mask is not None:
                sel = sel & mask.bool()
            logp = torch.log_softmax(out, dim=1)
            nll = -logp[sel, y[sel]].double().sum()
            hits = (out[sel].argmax(1) == y[sel]).sum().double()
            extra = torch.stack([nll, sel.sum().double(), hits])
            if grad_scale is None:
                out = None
            else:
                g = torch.zeros_like(out)
                g[sel] = (torch.softmax(out[sel], dim=1) - torch.nn.functional.one_hot(y[sel], out.size(1))) * grad_scale
                out = g
        if out_arg is not None:
            out_arg.copy_(out)
        if z_arg is not None:
            z_arg.copy_(z)
        return (out if want_out or ce is not None else out_arg), (z.contiguous() if want_z else z_arg), extra

    def run_rows(self, handle, x, lo, hi, out, kind, accumulate=False):
        return self.agg.run_rows(handle, x, lo, hi, out, kind, accumulate)

    def run(self, handle, x, kind):
        return self.agg.run(handle, x)

    def ce_stats_blocked(self, blk, bias, y, mask):
        return self.ce_stats(_rows_of(blk) + bias, y, mask)

    def rows_ok(self, handle):
        """RGBX_TEST_HUB_RANK = r: rank r behaves like one whose CSRs carry a hub-row plan (no row-range launches) —
        which ranks do depends on the graph, and the schedule's collectives must not."""
        hub = os.environ.get("RGBX_TEST_HUB_RANK")
        return hub is None or int(hub) != dist.get_rank()

    def gemm_tn(self, a, b, colsum=False, out=None, sums_out=None):
        res = a.t() @ b
        if out is not None:
            res = out.copy_(res)
        if not colsum:
            return res
        sums = a.sum(0)
        return res, (sums if sums_out is None else sums_out.copy_(sums))

    def blocked_to_rows(self, blk, bias=None):
        rows = _rows_of(blk).contiguous()
        return rows if bias is None else rows + bias

    def ce_stats(self, logits, y, mask):
        sel = (y >= 0) & (y < logits.size(1))
        if mask is not None:
            sel = sel & mask.bool()
        logp = torch.log_softmax(logits, dim=1)
        return torch.stack([-logp[sel, y[sel]].double().sum(), sel.sum().double(),
                            (logits[sel].argmax(1) == y[sel]).sum().double()])


def _stack_backend(self):
    return TorchStackBackend(self)


OracleAggregator.stack_backend = _stack_backend


def make_problem(n=97, e=900, f=12, c=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    loops = torch.randint(0, n, (9,), generator=g)
    ei = torch.cat([ei, torch.stack([loops, loops]), ei[:, :7]], dim=1)
    x = torch.randn(n, f, generator=g)
    y = torch.randint(0, c, (n,), generator=g)
    perm = torch.randperm(n, generator=g)
    masks = []
    a, b = int(0.6 * n), int(0.8 * n)
    for part in (perm[:a], perm[a:b], perm[b:]):
        m = torch.zeros(n, dtype=torch.bool)
        m[part] = True
        masks.append(m)
    return ei, x, y, masks


def leave_group():
    """End of a rank process of a test (its results are on disk by now): a barrier, so that no rank closes its sockets while a
    peer is still inside its own last collective; the process group destroyed; and the process left at once with status 0.
    Without the last step a rank now and then died in interpreter shutdown — "terminate called without an active exception",
    SIGABRT, with every assertion of the case already passed (round 5, under three pytest-xdist workers: a C++ thread of the
    backend still joinable when its owner is torn down). mp.spawn reads status 0 as a normal return."""
    import faulthandler
    import sys
    try:
        if dist.is_initialized():
            try:
                dist.barrier()
            except Exception:  # noqa: BLE001 - a peer that already failed: the spawn reports ITS error
                pass
            dist.destroy_process_group()
    finally:
        faulthandler.cancel_dump_traceback_later()
        sys.stdout.flush()
        sys.stderr.flush()
    os._exit(0)


def arm_deadline(seconds):
    """Hard per-case deadline of a rank process of a multi-rank GPU test: when it passes, the stacks of ALL threads of this
    rank go to stderr and the process exits (faulthandler, exit=True) — a stall fails the case within about a minute with
    the evidence, instead of holding the suite until its own limit (round 4: one case sat 5 min in a host-staged
    exchange). Every rank arms the same deadline, so the peers of a stuck rank end at the same time. Re-arming replaces the
    previous deadline. RGBX_TEST_DUMP_AFTER overrides the seconds."""
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("RGBX_TEST_DUMP_AFTER", seconds)), exit=True)


def disarm_deadline():
    import faulthandler
    faulthandler.cancel_dump_traceback_later()


def rccl_on_one_gpu_env(rank):
    """Environment of a rank that shares cuda:0 with the other ranks UNDER RCCL (rgb_experiment_amd/dist/sharing.py)."""
    from rgb_experiment_amd.dist.sharing import rccl_env
    return rccl_env(rank)


def _init(rank, world, port):
    """Process group of a test rank: gloo, or — RGBX_TEST_BACKEND=rccl, GPU tests only — RCCL with the ranks sharing cuda:0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if os.environ.get("RGBX_TEST_BACKEND") == "rccl":
        os.environ.update(rccl_on_one_gpu_env(rank))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)


def propagate_worker(rank, world, port, out_dir, exchange="halo"):
    """Forward + backward of the distributed propagate for every (loops_mode, kind)."""
    _init(rank, world, port)
    from rgb_experiment_amd.dist import Comm, DistGraph, partition_bounds
    ei, x, _, _ = make_problem()
    n = x.size(0)
    lo, hi = partition_bounds(n, world)[rank:rank + 2]
    go = torch.randn(n, x.size(1), generator=torch.Generator().manual_seed(5))
    res = {}
    for mode, kind in ((1, "gcn"), (2, "mean"), (0, "mean"), (0, "sum")):
        dg = DistGraph(ei, n, mode, Comm(), OracleAggregator(), exchange)
        res.setdefault("schemes", []).append(dg.scheme(x.size(1)))
        xl = x[lo:hi].clone().requires_grad_(True)
        out = dg.propagate(xl, kind)
        out.backward(go[lo:hi])
        res[f"{mode}_{kind}"] = (out.detach(), xl.grad)
    # K-step APPNP through ops.appnp_propagate (reshard: one pair of transposes around all K steps)
    from rgb_experiment_amd import ops
    dg = DistGraph(ei, n, 1, Comm(), OracleAggregator(), exchange)
    xl = x[lo:hi].clone().requires_grad_(True)
    out = ops.appnp_propagate(xl, dg, 4, 0.15)
    out.backward(go[lo:hi])
    res["appnp"] = (out.detach(), xl.grad)
    torch.save(res, os.path.join(out_dir, f"prop_{rank}.pt"))
    leave_group()


def plan_slices_worker(rank, world, port, out_dir):
    """Plans built from the ranks' slices of the edge list (plan.subsets_from_slices: bucketing of 1/P of the edges + one
    all-to-all of edge records per direction) against the plans every rank builds from the whole list: every tensor and every
    count of both halves, halo plans and every R x C grid of the world, all rewrite modes and weightings — bit for bit.
    Then the propagate itself through a DistGraph with the option on."""
    _init(rank, world, port)
    from rgb_experiment_amd.dist import Comm, DistGraph, partition_bounds
    from rgb_experiment_amd.dist import plan as P
    comm = Comm()
    checked = 0

    def same(a, b, where):
        nonlocal checked
        for k, v in vars(a).items():
            w = getattr(b, k)
            if torch.is_tensor(v):
                assert v.dtype == w.dtype and torch.equal(v, w), (where, k)
            else:
                assert v == w, (where, k, v, w)
            checked += 1

    for seed, (n, e) in enumerate([(103, 1200), (64, 40), (7, 300), (world, 3 * world)]):
        ei, _, _, _ = make_problem(n=n, e=e, f=4)
        g = torch.Generator().manual_seed(seed)
        ei = torch.cat([ei, torch.randint(0, n, (5,), generator=g).repeat(2, 1), ei[:, :9]], dim=1)  # self-loops, duplicates
        b = partition_bounds(n, world)
        for mode in (0, 1, 2):
            for group in [c for c in range(1, world + 1) if world % c == 0]:
                sub = P.subsets_from_slices(ei, n, mode, comm, group)
                glo, ghi = b[rank // group * group], b[(rank // group + 1) * group]
                ref = P.subsets_from_global(ei, n, mode, glo, ghi)
                assert (sub.lo, sub.hi, sub.nnz_total) == (glo, ghi, ref.nnz_total)
                assert torch.equal(sub.deg, ref.deg)
                for x, y in zip(sub.by_dst + sub.by_src, ref.by_dst + ref.by_src):
                    assert torch.equal(x, y), (n, mode, group)
                for kind in ("gcn", "mean", "sum"):
                    if group == 1:
                        whole = P.PartitionPlan(ei, n, world, rank, mode, kind)
                        mine = P.PartitionPlan.from_subsets(sub, n, world, rank, kind)
                        assert (whole.nnz_total, whole.nnz_local, whole.n_local) == (mine.nnz_total, mine.nnz_local, mine.n_local)
                        same(whole.fwd, mine.fwd, (n, mode, kind, "fwd"))
                        same(whole.bwd, mine.bwd, (n, mode, kind, "bwd"))
                    for pieces in (1, 3):
                        whole = P.GridPlan(ei, n, world, rank, mode, kind, group, pieces)
                        mine = P.GridPlan.from_subsets(sub, n, world, rank, kind, group, pieces)
                        assert whole.nnz_total == mine.nnz_total
                        same(whole.fwd, mine.fwd, (n, mode, kind, group, pieces, "fwd"))
                        same(whole.bwd, mine.bwd, (n, mode, kind, group, pieces, "bwd"))
    # the option on a DistGraph: same schemes, same rows, same gradients as with the option off
    ei, x, _, _ = make_problem()
    n = x.size(0)
    lo, hi = partition_bounds(n, world)[rank:rank + 2]
    go = torch.randn(n, x.size(1), generator=torch.Generator().manual_seed(5))
    exchanges = ["halo", "reshard", "auto"] + [f"{world // c}x{c}" for c in range(2, world) if world % c == 0]
    for exchange in exchanges:
        for mode, kind in ((1, "gcn"), (2, "mean"), (0, "sum")):
            got = []
            for on in (False, True):
                dg = DistGraph(ei, n, mode, comm, OracleAggregator(), exchange, plan_from_slices=on)
                xl = x[lo:hi].clone().requires_grad_(True)
                out = dg.propagate(xl, kind)
                out.backward(go[lo:hi])
                got.append((dg.scheme(x.size(1)), out.detach(), xl.grad))
                assert bool(dg._subsets) == on
            assert got[0][0] == got[1][0] and torch.equal(got[0][1], got[1][1]) and torch.equal(got[0][2], got[1][2]), exchange
    torch.save({"checked": checked}, os.path.join(out_dir, f"slices_{rank}.pt"))
    leave_group()


def reshard_chunk_worker(rank, world, port, out_dir, exchange="reshard"):
    """The column-shard propagate with the outgoing exchange in 1, 3 (ragged) and 4 pieces: identical rows."""
    _init(rank, world, port)
    from rgb_experiment_amd.dist import Comm, DistGraph, partition_bounds
    ei, x, _, _ = make_problem(n=103, e=1200, f=12)
    n = x.size(0)
    lo, hi = partition_bounds(n, world)[rank:rank + 2]
    outs = {}
    for chunks in (1, 3, 4):
        dg = DistGraph(ei, n, 1, Comm(), OracleAggregator(), exchange, pieces=chunks)
        assert dg.scheme(x.size(1)) == ("reshard" if exchange == "reshard" else "grid" + exchange)
        xl = x[lo:hi].clone().requires_grad_(True)
        out = dg.propagate(xl, "gcn")
        out.sum().backward()
        outs[chunks] = (out.detach(), xl.grad)
    torch.save(outs, os.path.join(out_dir, f"chunks_{rank}.pt"))
    leave_group()


def runner_worker(rank, world, port, out_dir, model_name, exchange="halo", interleave=True, release=False, fused=True,
                  pieces=None, cache=False, ahead=None):
    """Three epochs of DistRunner (train + evals) — compared by the test with single-process training.
    `ahead`: "all" = every epoch but the last announces a successor (epoch(more=True): the next training step is
    computed during this epoch's eval forwards); "stop" = so does the last one, and the loop then stops
    (discard_speculation)."""
    more = lambda last: bool(ahead) and (not last or ahead == "stop")
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    ei, x, y, masks = make_problem()
    torch.manual_seed(14530529)
    model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
    r = DistRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, comm=Comm(),
                   backend=OracleAggregator(), exchange=exchange, interleave_evals=interleave, fused=fused,
                   pieces=pieces, pieces_in=pieces or 1, cache_input_aggregate=cache, src_split=(pieces or 1) % 2 == 0)
    hist = [r.epoch(more=more(False))]
    if r.engine is not None:  # the module path's own structures (compared below) must exist before the release
        engine, r.engine = r.engine, None
        r.evaluate(1, sync=False)
        r.engine = engine
    if release:  # every structure exists after one epoch: the global edge list may go
        r.release_edge_list()
    hist += [r.epoch(more=more(False)), r.epoch(more=more(True))]
    if ahead:
        assert (r._spec is not None) == (ahead == "stop" and r.engine is not None)
    r.discard_speculation()
    both = None
    if r.engine is not None:  # the same weights through the fused schedule and through the modules
        eng = [r.engine.eval_stats(w) for w in (1, 2)]
        pair = r.engine.eval_pair(1, 2)  # the two forwards interleaved on one thread: the same bits
        assert all(torch.equal(a, b) for a, b in zip(eng, pair)), (eng, pair)
        engine, r.engine = r.engine, None
        both = (eng, [r.evaluate(w, sync=False)[0] for w in (1, 2)])
        r.engine = engine
    torch.save({"hist": hist, "logits_eval": r.logits(False), "lo": r.lo, "hi": r.hi, "engine": r.engine is not None,
                "eval_both": both,
                "state": {k: v.clone() for k, v in r.model.state_dict().items()}},
               os.path.join(out_dir, f"run_{model_name}_{rank}.pt"))
    leave_group()


def tasksplit_worker(rank, world, port, out_dir, model_name, exchange="reshard", stop_early=False):
    """Three epochs of dist.TaskSplitRunner: ranks [0, world / 2) train, the others evaluate; every epoch but the last
    announces a successor (`stop_early`: the last one does too, and the loop then stops — the step computed ahead
    must leave no trace)."""
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import TaskSplitRunner
    ei, x, y, masks = make_problem()
    torch.manual_seed(14530529)
    model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
    r = TaskSplitRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, backend=OracleAggregator(),
                        exchange=exchange)
    hist = [r.epoch(more=True), r.epoch(more=True), r.epoch(more=stop_early)]
    r.discard_speculation()
    torch.save({"hist": hist, "role": r.role, "lo": r.lo, "hi": r.hi,
                "state": {k: v.clone() for k, v in r.model.state_dict().items()}},
               os.path.join(out_dir, f"split_{model_name}_{rank}.pt"))
    leave_group()


def shared_eval_worker(rank, world, port, out_dir, model_name, exchange, fused, split):
    """The same three epochs twice: with the reference's two eval forwards per epoch and with share_eval_forward (one
    forward, both masks), on DistRunner (module route or fused schedule, the middle epoch announcing its successor) or on
    TaskSplitRunner. Saved: both histories, final states, exchanges and payload bytes per run."""
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner, TaskSplitRunner
    ei, x, y, masks = make_problem()
    res = {}
    for share in (False, True):
        torch.manual_seed(14530529)
        model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
        if split:
            r = TaskSplitRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, backend=OracleAggregator(),
                                exchange=exchange, share_eval_forward=share)
            comm = r.inner.comm
        else:
            comm = Comm()
            r = DistRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, comm=comm,
                           backend=OracleAggregator(), exchange=exchange, fused=fused, share_eval_forward=share)
        r.epoch()  # (builds every structure; its own exchanges are not counted)
        comm.exchanges, comm.bytes_sent = 0, 0
        hist = [r.epoch(more=True), r.epoch(more=False)]
        r.discard_speculation()
        res[share] = {"hist": hist, "exchanges": comm.exchanges, "bytes": comm.bytes_sent,
                      "engine": getattr(r, "engine", None) is not None, "role": getattr(r, "role", None),
                      "state": {k: v.clone() for k, v in r.model.state_dict().items()}}
        dist.barrier()
    torch.save(res, os.path.join(out_dir, f"shared_{model_name}_{rank}.pt"))
    leave_group()


def failing_eval_worker(rank, world, port, out_dir):
    """Rank 1 raises inside an eval forward of the second epoch (the interleaved pair, on a helper thread): the
    process must END (Comm.abort), so that the job terminates instead of leaving rank 0 in an all-to-all."""
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner

    class Failing(OracleAggregator):
        armed = False

        def run(self, handle, x, y=None, kind=None):
            import threading
            if Failing.armed and threading.current_thread() is not threading.main_thread():  # an eval thread
                raise RuntimeError("injected: aggregation failed in an eval forward")
            return super().run(handle, x, y, kind)

    ei, x, y, masks = make_problem()
    torch.manual_seed(14530529)
    model = build_model(M, "gcn", x.size(1), int(y.max()) + 1)
    r = DistRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, comm=Comm(),
                   backend=Failing(), exchange="reshard", interleave_evals=True)
    r.epoch()  # the first epoch's evals run one after the other on the main thread
    open(os.path.join(out_dir, f"first_epoch_done_{rank}"), "w").close()
    Failing.armed = rank == 1
    r.epoch()
    open(os.path.join(out_dir, f"second_epoch_done_{rank}"), "w").close()  # must not be reached by rank 1
    leave_group()


def exchange_count_worker(rank, world, port, out_dir, model_name, resident):
    """Counts the row exchanges of one steady-state epoch, with and without resident input features."""
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner

    class CountingComm(Comm):
        calls = 0

        def all_to_all_rows(self, send, send_counts, recv_counts, tag=None):
            CountingComm.calls += 1
            return super().all_to_all_rows(send, send_counts, recv_counts, tag)

    ei, x, y, masks = make_problem()
    torch.manual_seed(14530529)
    model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
    r = DistRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, comm=CountingComm(),
                   backend=OracleAggregator(), exchange="halo", resident_features=resident)
    hist = [r.epoch()]
    CountingComm.calls = 0
    hist.append(r.epoch())
    torch.save({"hist": hist, "exchanges": CountingComm.calls},
               os.path.join(out_dir, f"cnt_{model_name}_{int(resident)}_{rank}.pt"))
    leave_group()


def experiment_worker(rank, world, port, out_dir, model_name, on_gpu=False, classes=None):
    """experiment() called as one of `world` ranks (environment as torch.distributed.run sets it): the distributed
    route of rgb_experiment_amd.itexperiments.experiment."""
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "RGBX_DIST_BACKEND": "gloo"})
    if on_gpu:
        arm_deadline(120)
    if on_gpu and os.environ.get("RGBX_TEST_BACKEND") == "rccl":
        # the product's own backend, the ranks sharing cuda:0: experiment() sees WORLD_SIZE > visible GPUs and prepares RCCL for
        # it by itself (dist/sharing.py) - nothing but the backend's name is set here
        os.environ["RGBX_DIST_BACKEND"] = "nccl"
    if not on_gpu:
        os.environ["RGBX_TEST_AGGREGATOR"] = "_dist_worker:OracleAggregator"
    torch.set_num_threads(1)
    import rgb_experiment_amd as R
    n, e, f, c = (5000, 60000, 32, 8) if on_gpu else (97, 900, 12, 5)
    ei, x, y, masks = make_problem(n=n, e=e, f=f, c=classes or c)
    data = R.Data(x=x, y=y, edge_index=ei)
    data.train_mask, data.val_mask, data.test_mask = masks
    params = R.InitialParameters.defaults_for(model_name)
    params["hidden_unit"] = 32
    res = R.experiment(params, specify_data=True, data=data, model_name=model_name, learning_rate=0.01, epoch=6,
                       need_to_reappear=True, print_print=False, return_model=True, need_all_metrics=True,
                       keep_valid_data_mask=True)
    torch.save({"metrics": {k: res[k] for k in ("ACC", "precision_score", "recall_score", "f1_macro", "f1_micro")},
                "history": res["history"], "distributed": res["distributed"],
                "state": {k: v.cpu().clone() for k, v in res["model"].state_dict().items()}},
               os.path.join(out_dir, f"exp_{model_name}_{world}_{rank}.pt"))
    leave_group()


def build_model(M, name, f, c):
    if name.endswith("_wide"):  # in <= hidden, hidden % 32 == 0: the first conv takes the fused / resident route
        cls = {"gcn_wide": M.GCN, "graphsage_wide": M.GraphSAGE, "graphsage2_wide": M.GraphSAGE2}[name]
        return cls(num_layers=2, hidden_unit=64, input_dim=f, output_dim=c, dropout_rate=0.5)
    if name.endswith("_grid"):  # widths the fused per-rank schedule takes (dist/stack.py): hidden 96, 32 logits
        cls, layers = {"gcn_grid": (M.GCN, 2), "gcn3_grid": (M.GCN, 3), "graphsage_grid": (M.GraphSAGE, 2),
                       "graphsage2_grid": (M.GraphSAGE2, 3)}[name]
        return cls(num_layers=layers, hidden_unit=96, input_dim=f, output_dim=32, dropout_rate=0.5)
    if name == "gcn_bench":  # the models bench.py times (in = hidden = classes = 128)
        return M.GCN(num_layers=2, hidden_unit=128, input_dim=f, output_dim=c, dropout_rate=0.5)
    if name == "graphsage_bench":
        return M.GraphSAGE(num_layers=2, hidden_unit=128, input_dim=f, output_dim=c, dropout_rate=0.5)
    if name == "appnpstack_bench":
        return M.APPNPStack(hidden_unit=64, input_dim=f, output_dim=c, K=10, alpha=0.1, dropout_rate=0.5)
    if name == "gcn":
        return M.GCN(num_layers=3, hidden_unit=16, input_dim=f, output_dim=c, dropout_rate=0.5)
    if name == "graphsage":
        return M.GraphSAGE(num_layers=2, hidden_unit=16, input_dim=f, output_dim=c, dropout_rate=0.5)
    if name == "graphsage2":
        return M.GraphSAGE2(num_layers=2, hidden_unit=16, input_dim=f, output_dim=c, dropout_rate=0.5)
    if name == "appnpstack":
        return M.APPNPStack(hidden_unit=16, input_dim=f, output_dim=c, K=4, alpha=0.1, dropout_rate=0.5)
    if name == "gat":
        return M.GAT(num_layers=2, hidden_unit=4, heads=3, input_dim=f, output_dim=c, dropout_rate=0.5)
    raise KeyError(name)


def lopsided_problem():
    """make_problem's sizes with every edge inside the first 40 % of the nodes: under a 1-D partition over 2-4 ranks the last
    rank(s) own rows WITHOUT any edge (only the self-loops a GCN adds), send nothing and receive nothing in every exchange —
    empty entries of the grouped send / recv lists, empty halo plans, zero-row pieces."""
    ei, x, y, masks = make_problem(n=5000, e=60000, f=32, c=8)
    return ei % 2000, x, y, masks


def problem_of(model_name):
    """'gcn@lopsided' -> ('gcn', the lopsided problem), 'gcn@hub' -> hub_problem(); plain names -> make_problem(n=5000, ...)."""
    base, _, variant = model_name.partition("@")
    if variant == "lopsided":
        return base, lopsided_problem()
    if variant == "hub":
        return base, hub_problem()
    return base, make_problem(n=5000, e=60000, f=32, c=8)


def bench_problem_S():
    """bench.py's workload S (BASELINE configs[1]: |V| = 200 k, |E| = 4 M, d = 128, 128 classes; same seeds)."""
    n, e = 200_000, 4_000_000
    ei = torch.randint(0, n, (2, e), generator=torch.Generator().manual_seed(1234567), dtype=torch.int64)
    x = torch.randn(n, 128, generator=torch.Generator().manual_seed(1234568))
    y = torch.randint(0, 128, (n,), generator=torch.Generator().manual_seed(1234569))
    k = torch.arange(n) % 5
    return ei, x, y, [k < 3, k == 3, k == 4]


def hub_problem():
    """make_problem(n=5000, ...) plus two hub nodes: node 7 with 3000 extra in-edges, node 11 with 3000 extra
    out-edges — rows beyond LONG_ROW_SLOTS in the forward and in the transposed CSR (hub-row plans)."""
    ei, x, y, masks = make_problem(n=5000, e=60000, f=32, c=8)
    g = torch.Generator().manual_seed(99)
    others = torch.randint(0, 5000, (3000,), generator=g)
    extra = torch.cat([torch.stack([others, torch.full_like(others, 7)]),
                       torch.stack([torch.full_like(others, 11), others])], dim=1)
    return torch.cat([ei, extra], dim=1), x, y, masks


def gpu_runner_worker(rank, world, port, out_dir, model_name, exchange="halo", hub=False, ahead=True, size=None):
    """Rehearsal of the real per-rank HIP path: `world` ranks share cuda:0, collectives go through gloo
    with host staging (RCCL cannot put two ranks on one device). `ahead`: the first epoch announces the second
    (epoch(more=True): the fused schedule computes the second training step beside the first epoch's eval forwards)."""
    arm_deadline(150 if size == "S" else 90)  # a stuck rank ends with the stacks of all its threads on stderr
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    dev = torch.device("cuda:0")
    ei, x, y, masks = bench_problem_S() if size == "S" else hub_problem() if hub else make_problem(n=5000, e=60000, f=32, c=8)
    torch.manual_seed(14530529)
    model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
    r = DistRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, comm=Comm(), exchange=exchange)
    hist = [r.epoch(more=ahead), r.epoch()]
    torch.cuda.synchronize()
    torch.save({"hist": hist, "logits_train": r.logits(True).cpu(), "lo": r.lo, "hi": r.hi, "backend": dist.get_backend(),
                "engine": r.engine is not None, "state": {k: v.cpu() for k, v in r.model.state_dict().items()}},
               os.path.join(out_dir, f"gpu_{model_name}_{rank}.pt"))
    leave_group()
    disarm_deadline()


def gpu_plan_slices_worker(rank, world, port, out_dir, model_name, exchange):
    """Two epochs of the per-rank HIP path with the plans built from the whole edge list, then again (same seeds) with
    RGBX_PLAN_FROM_SLICES=1: the edge records travel as int64 rows through the run's own backend (RCCL with the ranks sharing
    the GPU, or gloo with host staging)."""
    arm_deadline(120)
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    from rgb_experiment_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    ei, x, y, masks = hub_problem()
    out = {}
    for on in (False, True):
        os.environ["RGBX_PLAN_FROM_SLICES"] = "1" if on else "0"
        clear_cache()
        torch.manual_seed(14530529)
        model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
        r = DistRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, comm=Comm(), exchange=exchange)
        hist = [r.epoch(more=True), r.epoch()]
        torch.cuda.synchronize()
        used = any(bool(g._subsets) for g in r.graphs.values()) if hasattr(r, "graphs") else None
        out[on] = {"hist": hist, "logits": r.logits(True).cpu(), "used": used,
                   "state": {k: v.cpu() for k, v in r.model.state_dict().items()}}
    os.environ.pop("RGBX_PLAN_FROM_SLICES", None)
    torch.save(out, os.path.join(out_dir, f"gpuslices_{model_name}_{rank}.pt"))
    leave_group()
    disarm_deadline()


def gpu_tasksplit_worker(rank, world, port, out_dir, model_name, exchange="reshard"):
    """dist.TaskSplitRunner on the real kernels: `world` ranks share cuda:0 (gloo staging), two epochs, the first one
    announcing the second (the training group computes the second step ahead)."""
    arm_deadline(90)
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import TaskSplitRunner
    dev = torch.device("cuda:0")
    ei, x, y, masks = make_problem(n=5000, e=60000, f=32, c=8)
    torch.manual_seed(14530529)
    model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
    r = TaskSplitRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, exchange=exchange)
    hist = [r.epoch(more=True), r.epoch()]
    torch.cuda.synchronize()
    torch.save({"hist": hist, "role": r.role, "logits_train": r.logits(True).cpu(), "lo": r.lo, "hi": r.hi,
                "state": {k: v.cpu() for k, v in r.model.state_dict().items()}},
               os.path.join(out_dir, f"gpusplit_{model_name}_{rank}.pt"))
    leave_group()
    disarm_deadline()


def comm_selftest_worker(rank, world, port, out_dir, sabotage=False):
    """Comm.self_test_views (view exchange with aliased / empty entries + several works in flight waited out of order)."""
    _init(rank, world, port)
    from rgb_experiment_amd.dist import Comm
    comm = Comm()
    if sabotage and rank == 1:  # a backend that pairs the second piece's views wrongly: the self-test must say no
        real = comm.all_to_all_views

        def wrong(send, recv, tag=None):
            work = real(send, recv, tag)
            if tag == "self-test piece 1":
                for t in recv:
                    if t.numel():
                        t.add_(1.0)
            return work
        comm.all_to_all_views = wrong
    ok = comm.self_test_views(torch.device("cpu"))
    torch.save({"ok": ok, "exchanges": comm.exchanges}, os.path.join(out_dir, f"selftest_{rank}.pt"))
    leave_group()


def gpu_runner_worker_multi(rank, world, port, out_dir, cases):
    """gpu_runner_worker for SEVERAL (model_name, exchange) cases in one set of rank processes (one interpreter start-up, one
    process group): every case builds its own model and DistRunner, runs its two epochs, saves
    gpu_<model>_<exchange>_<rank>.pt, and leaves nothing behind for the next one (graph caches dropped)."""
    arm_deadline(120)  # imports, process group, the first case
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    from rgb_experiment_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    for i, (model_name, exchange) in enumerate(cases):
        if i:
            arm_deadline(60)  # per case from here on
        base, (ei, x, y, masks) = problem_of(model_name)
        torch.manual_seed(14530529)
        model = build_model(M, base, x.size(1), int(y.max()) + 1)
        r = DistRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, comm=Comm(), exchange=exchange)
        hist = [r.epoch(more=True), r.epoch()]
        torch.cuda.synchronize()
        torch.save({"hist": hist, "logits_train": r.logits(True).cpu(), "lo": r.lo, "hi": r.hi,
                    "engine": r.engine is not None, "backend": dist.get_backend()},
                   os.path.join(out_dir, f"gpu_{model_name}_{exchange}_{rank}.pt"))
        del r, model
        clear_cache()
        torch.cuda.empty_cache()
        dist.barrier()
    leave_group()
    disarm_deadline()


def single_gpu_worker(rank, out_path, jobs):
    """The ONE-GPU reference runs of tests/test_gpu_dist.py, in a process of their own: the test's parent process then never
    holds a GPU context, so a set of N ranks sharing the card is N processes on it, not N + 1 (beyond the hardware's queue
    slots the driver time-slices the queues of processes that wait on one another through gloo: the same S-size case took
    15 s on one box and 96 s on another, and minutes with side streams on top). jobs = [(key, model_name, hub, size)]."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    out = {}
    nll = torch.nn.functional.nll_loss
    for key, model_name, hub, size in jobs:
        base, small = problem_of(model_name)
        ei, x, y, masks = bench_problem_S() if size == "S" else hub_problem() if hub else small
        torch.manual_seed(14530529)
        model = build_model(M, base, x.size(1), int(y.max()) + 1).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        ei, x, y = ei.to(dev), x.to(dev), y.to(dev)
        masks = [m.to(dev) for m in masks]
        hist = []
        for _ in range(2):
            model.train()
            opt.zero_grad()
            o = model(x, ei)["out"]
            loss = nll(o[masks[0]], y[masks[0]])
            loss.backward()
            opt.step()
            model.eval()
            with torch.no_grad():
                ev = model(x, ei)["out"]
            hist.append((loss.item(), nll(ev[masks[1]], y[masks[1]]).item(), nll(ev[masks[2]], y[masks[2]]).item()))
        model.train()
        with torch.no_grad():
            out[key] = (hist, model(x, ei)["emb"].cpu())
        del model, opt
        clear_cache()
    torch.save(out, out_path)


def experiment_single_worker(rank, out_path, model_names):
    """experiment() on one GPU with the data / arguments of experiment_worker(on_gpu=True), for every model of `model_names`,
    in ONE process of its own (a process per model was 3 s of interpreter start-up and HIP context each)."""
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RGBX_DIST_BACKEND"):
        os.environ.pop(key, None)
    import rgb_experiment_amd as R
    out = {}
    for model_name in model_names:
        ei, x, y, masks = make_problem(n=5000, e=60000, f=32, c=8)
        data = R.Data(x=x, y=y, edge_index=ei)
        data.train_mask, data.val_mask, data.test_mask = masks
        params = R.InitialParameters.defaults_for(model_name)
        params["hidden_unit"] = 32
        one = R.experiment(params, specify_data=True, data=data, model_name=model_name, learning_rate=0.01, epoch=6,
                           need_to_reappear=True, print_print=False, return_model=True, need_all_metrics=True,
                           keep_valid_data_mask=True, use_hip_graph=False)
        out[model_name] = {"history": one["history"], "ACC": one["ACC"],
                           "state": {k: v.cpu() for k, v in one["model"].state_dict().items()}}
    torch.save(out, out_path)


def gpu_ahead_pair_worker(rank, world, port, out_dir, model_name, exchange, size):
    """The same two epochs twice in one set of rank processes: with the second training step computed ahead of the first
    epoch's eval forwards (epoch(more=True)) and without. Saved per variant: history, train-mode logits, state."""
    arm_deadline(180 if size == "S" else 120)
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    from rgb_experiment_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    ei, x, y, masks = bench_problem_S() if size == "S" else make_problem(n=5000, e=60000, f=32, c=8)
    res = {}
    for ahead in (True, False):
        torch.manual_seed(14530529)
        model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
        r = DistRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, comm=Comm(), exchange=exchange)
        hist = [r.epoch(more=ahead), r.epoch()]
        torch.cuda.synchronize()
        res[ahead] = {"hist": hist, "logits_train": r.logits(True).cpu(), "engine": r.engine is not None,
                      "state": {k: v.cpu() for k, v in r.model.state_dict().items()}}
        del r, model
        clear_cache()
        torch.cuda.empty_cache()
        dist.barrier()
    torch.save(res, os.path.join(out_dir, f"ahead_{model_name}_{rank}.pt"))
    leave_group()
    disarm_deadline()


def gpu_shared_eval_worker(rank, world, port, out_dir, model_name, exchange, size):
    """share_eval_forward on the real kernels (the ranks sharing cuda:0): the reference's two eval forwards per epoch
    against one forward serving both masks, two epochs each (the first announcing the second). Saved: histories, states,
    exchanges and payload bytes of the two epochs."""
    arm_deadline(180 if size == "S" else 120)
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    from rgb_experiment_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    ei, x, y, masks = bench_problem_S() if size == "S" else make_problem(n=5000, e=60000, f=32, c=8)
    res = {}
    for share in (False, True):
        torch.manual_seed(14530529)
        model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
        comm = Comm()
        r = DistRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, comm=comm, exchange=exchange,
                       share_eval_forward=share)
        comm.exchanges, comm.bytes_sent = 0, 0
        hist = [r.epoch(more=True), r.epoch()]
        torch.cuda.synchronize()
        res[share] = {"hist": hist, "exchanges": comm.exchanges, "bytes": comm.bytes_sent, "engine": r.engine is not None,
                      "state": {k: v.cpu() for k, v in r.model.state_dict().items()}}
        del r, model
        clear_cache()
        torch.cuda.empty_cache()
        dist.barrier()
    torch.save(res, os.path.join(out_dir, f"shared_{model_name}_{rank}.pt"))
    leave_group()
    disarm_deadline()


def rccl_probe_worker(rank, world, port, out_dir):
    """Does RCCL come up with `world` ranks on cuda:0 on this box (dist/sharing.py)? One all-reduce and one grouped
    send / recv list with an empty entry; a file per rank says yes."""
    os.environ["RGBX_TEST_BACKEND"] = "rccl"
    _init(rank, world, port)
    dev = torch.device("cuda", 0)
    t = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    send = [torch.full((4, 8), float(10 * rank + q), device=dev) if q != rank else torch.empty((0, 8), device=dev)
            for q in range(world)]
    recv = [torch.empty((4, 8), device=dev) if q != rank else torch.empty((0, 8), device=dev) for q in range(world)]
    dist.all_to_all(recv, send, async_op=True).wait()
    torch.cuda.synchronize()
    ok = t[0].item() == world * (world + 1) / 2 and all(bool((recv[q] == 10 * q + rank).all()) for q in range(world) if q != rank)
    if ok:
        torch.save({"backend": dist.get_backend(), "nccl": torch.cuda.nccl.version()}, os.path.join(out_dir, f"probe_{rank}.pt"))
    leave_group()


def gpu_interleave_worker(rank, world, port, out_dir, model_name, exchange, size):
    """The same three epochs twice in one set of ranks: val and test forwards issued by two host threads on two HIP streams
    (DistRunner's default from the second epoch on) and one after the other on the caller's stream. Saved: both histories and
    both sets of train-mode logits."""
    os.environ["RGBX_INTERLEAVE"] = "always"  # no host-bound verdict in between (DistRunner._settle_interleave)
    arm_deadline(150 if size == "S" else 90)
    _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    from rgb_experiment_amd.graph import clear_cache
    dev = torch.device("cuda:0")
    ei, x, y, masks = bench_problem_S() if size == "S" else make_problem(n=5000, e=60000, f=32, c=8)
    res = {"backend": dist.get_backend()}
    for tag, inter in (("interleaved", True), ("sequential", False)):
        torch.manual_seed(14530529)
        model = build_model(M, model_name, x.size(1), int(y.max()) + 1)
        r = DistRunner(model, ei, x, y, masks, rank, world, dev, lr=0.01, comm=Comm(), exchange=exchange,
                       interleave_evals=inter, fused=False)
        hist = [r.epoch(), r.epoch(), r.epoch()]
        torch.cuda.synchronize()
        res[tag] = {"hist": hist, "logits": r.logits(True).cpu(), "threads": bool(r.interleave_evals and r.engine is None)}
        del r, model
        clear_cache()
        dist.barrier()
    torch.save(res, os.path.join(out_dir, f"inter_{model_name}_{rank}.pt"))
    leave_group()
    disarm_deadline()


def identity_exchange_worker(rank, world, port, out_dir, idents):
    """dist/sharing.py's store exchange as init_rccl does it (env:// rendezvous, identities through the store, then a process
    group made FROM that store) — on gloo, with made-up identities."""
    import datetime
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from rgb_experiment_amd.dist import sharing
    store, r, w = sharing.rendezvous_store(120)
    got = sharing.exchange_identities(store, r, w, idents[rank])
    dist.init_process_group("gloo", store=store, rank=r, world_size=w, timeout=datetime.timedelta(seconds=120))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    torch.save({"got": got, "shared": sharing.share_a_device(got), "sum": t.item(), "rank": r, "world": w},
               os.path.join(out_dir, f"ident_{rank}.pt"))
    leave_group()


def propagate_fuzz_worker(rank, world, port, out_dir, seeds, own_group=True):
    """Distributed propagate (every loops mode / kind) + K-step APPNP, forward and backward, on RANDOM problems: node counts from
    `world` itself (one row per rank) upward, edge lists from empty to hub-heavy (tests/test_gpu_fuzz.make_graph), widths that do
    and do not divide by the world size or the grid's column count (the scheme then falls back to halo for that width), every
    exchange scheme and piece count. Every rank computes the single-process oracle itself and compares its own rows."""
    import random

    import test_gpu_fuzz as F
    from oracle import ref_cpu as O
    if own_group:
        _init(rank, world, port)
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.dist import Comm, DistGraph, partition_bounds
    grids = [f"{r}x{world // r}" for r in range(1, world) if world % r == 0 and world // r > 1]
    bad = []
    for seed in seeds:
        rng = random.Random(seed)
        n = rng.choice([world, world + 1, 2 * world + 1, 17, 64, 97, 300, 1000])
        n = max(n, world)
        f = rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 30, 32, 48])
        exchange = rng.choice(["halo", "reshard", "auto"] + grids)
        pieces = rng.choice([None, 1, 2, 3, 4])
        ei = F.make_graph(rng, n)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(n, f, generator=g)
        go = torch.randn(n, f, generator=g)
        lo, hi = partition_bounds(n, world)[rank:rank + 2]
        desc = f"seed={seed} world={world} n={n} E={ei.size(1)} f={f} exchange={exchange} pieces={pieces}"

        def chk(ok, what):  # (recorded, never raised: a rank that left a case early would miss its peers' next collective)
            if not ok:
                bad.append((desc, repr(what)))
        if True:
            for mode, kind in ((1, "gcn"), (2, "mean"), (0, "mean"), (0, "sum")):
                dg = DistGraph(ei, n, mode, Comm(), OracleAggregator(), exchange, pieces=pieces)
                xl = x[lo:hi].clone().requires_grad_(True)
                out = dg.propagate(xl, kind)
                out.backward(go[lo:hi])
                rei, _ = O.rewrite_edges(ei, n, mode)
                xr = x.double().requires_grad_(True)  # the oracle in float64: thousands of duplicate edges between a handful
                if kind == "gcn":                       # of nodes are summed one by one by its index_add_ (2e-4 off in float32)
                    _, w = O.gcn_norm(ei, None, n)
                    want = O.propagate(rei, xr, n, w.double(), "add")
                else:
                    want = O.propagate(rei, xr, n, None, "add" if kind == "sum" else "mean")
                want.backward(go.double())
                tol = 1e-4 * max(1.0, want.detach().abs().max().item(), xr.grad.abs().max().item())  # (the CPU aggregator of this rehearsal sums in float32, one term at a time)
                if hi > lo:
                    chk((out.detach().double() - want.detach()[lo:hi]).abs().max().item() <= tol, (mode, kind, "out"))
                    chk((xl.grad.double() - xr.grad[lo:hi]).abs().max().item() <= tol, (mode, kind, "grad"))
            K, alpha = rng.choice([1, 2, 4]), rng.choice([0.0, 0.15])
            dg = DistGraph(ei, n, 1, Comm(), OracleAggregator(), exchange, pieces=pieces)
            xl = x[lo:hi].clone().requires_grad_(True)
            out = ops.appnp_propagate(xl, dg, K, alpha)
            out.backward(go[lo:hi])
            xr = x.double().requires_grad_(True)
            want = O.appnp(xr, ei, K, alpha)
            want.backward(go.double())
            tol = 1e-4 * max(1.0, want.detach().abs().max().item(), xr.grad.abs().max().item())
            if hi > lo:
                chk((out.detach().double() - want.detach()[lo:hi]).abs().max().item() <= tol, ("appnp out", K, alpha))
                chk((xl.grad.double() - xr.grad[lo:hi]).abs().max().item() <= tol, ("appnp grad", K, alpha))
        # (an exception ends this rank; its peers then fail in the next collective and the spawn reports it)
    torch.save(bad, os.path.join(out_dir, f"propfuzz_{rank}.pt"))
    if own_group:
        leave_group()


def runner_fuzz_worker(rank, world, port, out_dir, seeds, own_group=True):
    """DistRunner (fused per-rank schedule where the widths allow it, the modules otherwise) for two epochs on RANDOM problems
    and models, against the same two epochs of single-process oracle training that every rank runs for itself: train losses
    (the second one sees the first update: gradients, their all-reduce and Adam are in it), eval losses and hit counts."""
    import random

    import test_gpu_fuzz as F
    from oracle import ref_cpu as O
    if own_group:
        _init(rank, world, port)
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import Comm, DistRunner
    from rgb_experiment_amd.graph import clear_cache
    grids = [f"{r}x{world // r}" for r in range(1, world) if world % r == 0 and world // r > 1]
    nll = torch.nn.functional.nll_loss
    bad = []
    for seed in seeds:
        rng = random.Random(seed)
        n = rng.choice([4 * world + 1, 97, 150, 400])
        f = rng.choice([8, 12, 16, 24, 32, 48])
        c = rng.choice([4, 8, 24, 32])
        hidden = rng.choice([8, 16, 24, 32, 48, 96])
        layers = rng.choice([2, 2, 3])
        kind = rng.choice(["gcn", "gcn", "graphsage", "graphsage", "graphsage2", "appnpstack", "gat"])
        exchange = rng.choice(["halo", "reshard", "auto", "replicate"] + grids)
        pieces = rng.choice([None, 1, 2, 3])
        ei = F.make_graph(rng, n)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(n, f, generator=g)
        y = torch.randint(0, c, (n,), generator=g)
        r_ = torch.rand(n, generator=g)
        masks = [r_ < 0.5, (r_ >= 0.5) & (r_ < 0.75), r_ >= 0.75]
        masks[0][0] = True  # (val / test may stay empty on the smallest graphs: nan there, compared as "no rows")
        desc = f"seed={seed} world={world} {kind} n={n} E={ei.size(1)} f={f} hidden={hidden} classes={c} layers={layers} exchange={exchange} pieces={pieces}"
        torch.manual_seed(seed)
        if kind == "gcn":
            model = M.GCN(num_layers=layers, hidden_unit=hidden, input_dim=f, output_dim=c, dropout_rate=0.5)
            fwd = lambda p, tr: O.gcn_forward(p, x, ei, layers, tr)
        elif kind == "graphsage":
            model = M.GraphSAGE(num_layers=layers, hidden_unit=hidden, input_dim=f, output_dim=c, dropout_rate=0.5)
            fwd = lambda p, tr: O.graphsage_forward(p, x, ei, layers, tr)
        elif kind == "graphsage2":
            model = M.GraphSAGE2(num_layers=layers, hidden_unit=hidden, input_dim=f, output_dim=c, dropout_rate=0.5)
            fwd = lambda p, tr: O.graphsage2_forward(p, x, ei, layers, tr)
        elif kind == "gat":
            heads = rng.choice([1, 2, 4])
            model = M.GAT(num_layers=layers, hidden_unit=max(hidden // 8, 1), heads=heads, input_dim=f, output_dim=c, dropout_rate=0.5)
            fwd = lambda p, tr: O.gat_forward(p, x, ei, layers, heads, tr)
        else:
            K = rng.choice([1, 3])
            model = M.APPNPStack(hidden_unit=hidden, input_dim=f, output_dim=c, K=K, alpha=0.1, dropout_rate=0.5)
            fwd = lambda p, tr: O.appnp_stack_forward(p, x, ei, K, 0.1, tr)
        if exchange == "replicate" and (kind in ("gat", "appnpstack") or f > hidden):
            exchange = "auto"  # (the replicate scheme serves conv stacks whose first layer aggregates before it transforms)
        params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()
                  if "lin_dst" not in k}
        trainable = [k for k, _ in model.named_parameters() if "lin_dst" not in k]
        opt = torch.optim.Adam([params[k] for k in trainable], lr=0.01)
        want, g_first = [], None
        for _ in range(2):
            opt.zero_grad()
            out = fwd(params, True)["out"]
            loss = nll(out[masks[0]], y[masks[0]])
            loss.backward()
            if g_first is None:  # the first backward's gradients: what the distributed backward is held to, tightly (below)
                g_first = {k: params[k].grad.detach().clone() for k in trainable}
            opt.step()
            with torch.no_grad():
                ev = fwd(params, False)["out"]
            want.append((loss.item(),
                         [nll(ev[m], y[m]).item() if int(m.sum()) else 0.0 for m in masks[1:]],
                         [int((ev[m].argmax(1) == y[m]).sum()) for m in masks[1:]]))
        clear_cache()
        if "desc" in os.environ.get("RGBX_FUZZ_SHOW", ""):
            print(f"[rank {rank}] {desc}", flush=True)
        # the epoch split by task on an even world (half the ranks train, half evaluate; 2 ranks: each on the whole graph)
        split = world % 2 == 0 and exchange != "replicate" and rng.random() < 0.3
        if split and "x" in exchange:  # a grid must factor the GROUP (TaskSplitRunner refuses anything else by name)
            half = world // 2
            fits = [f"{a}x{half // a}" for a in range(1, half) if half % a == 0 and half // a > 1]
            exchange = rng.choice(fits) if fits else "auto"
        try:
            if split:
                from rgb_experiment_amd.dist import TaskSplitRunner
                r = TaskSplitRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01,
                                    backend=OracleAggregator(), exchange=exchange)
                desc += " task-split"
            else:
                r = DistRunner(model, ei, x, y, masks, rank, world, torch.device("cpu"), lr=0.01, comm=Comm(),
                               backend=OracleAggregator(), exchange=exchange, pieces=pieces, pieces_in=pieces or 1)
            #  the distributed BACKWARD on its own, before any optimizer step: halo / transposed exchanges of
            # the gradient rows, the all-reduce of the parameter gradients — held to the oracle's first-step gradients at the
            # single-GPU gradient tests' bound, where the loss after an Adam step (below) can only take a gross-error bound.
            # The probe's forward moves BatchNorm's running statistics: the state is put back before the epochs run.
            import copy
            inner = r.inner if split else r
            if not split or r.role == "train":
                keep = copy.deepcopy(inner.model.state_dict())
                inner._forward_backward()
                inner._sync_grads()
                from oracle import large as OL
                got = {k: p.grad.detach().clone() for k, p in inner.model.named_parameters() if "lin_dst" not in k and p.grad is not None}
                rep = OL.compare_grads(got, {k: g_first[k] for k in got})
                if rep["max_rel"] > float(os.environ.get("RGBX_FUZZ_GRAD_REL", "2e-3")) and rep["max_vs_bound"] > 1e-5:
                    bad.append((desc, f"first backward: gradient of {rep['worst']} off by {rep['max_rel']:.2e} of its scale "
                                      f"({rep['max_abs']:.2e} absolute)", getattr(r, "engine", None) is not None))
                if "grad" in os.environ.get("RGBX_FUZZ_SHOW", ""):
                    print(f"[rank {rank}] {desc}: first-backward gradients max_rel {rep['max_rel']:.2e} "
                          f"max_abs {rep['max_abs']:.2e} ({rep['worst']})", flush=True)
                inner.model.load_state_dict(keep)
                inner.opt.zero_grad(set_to_none=False)
                if getattr(inner, "engine", None) is not None:
                    inner.engine.discard_speculation()
            hist = [r.epoch(more=True), r.epoch()]
            if split:
                r.discard_speculation()
        except (NotImplementedError, ValueError) as exc:  # a combination the runner refuses by name, on every rank alike
            bad.append((desc, "refused: " + repr(exc)[:160])) if "refus" in os.environ.get("RGBX_FUZZ_SHOW", "") else None
            continue
        for step in range(2):
            tl, wl = hist[step][0], want[step][0]
            # the first loss is a pure forward; the second has ONE Adam step in it, and Adam turns the rounding noise of a
            # parameter whose true gradient is ~0 (a bias in front of a BatchNorm, GAT's attention vectors once a hub has made
            # the rows alike) into a +-lr step: two correct runs differ by O(lr) there (soak, round 4: 2 of 4,000 cases at
            # 1e-4 and 3e-3 relative, a 17-node and a 33-node hub graph) — tight on the first, a gross-error bound on the second
            if abs(tl - wl) > (5e-5 if step == 0 else 1e-2) * max(1.0, abs(wl)):
                bad.append((desc, f"train loss of epoch {step}: {tl} vs {wl}", getattr(r, "engine", None) is not None))
        # eval numbers of the last epoch against the oracle's eval forward on THIS run's trained weights (two separately
        # trained runs differ by Adam's +-lr steps on the biases in front of a BatchNorm, whose true gradient is zero)
        own = {k: v.detach().clone() for k, v in r.model.state_dict().items() if "lin_dst" not in k}
        with torch.no_grad():
            ev = fwd(own, False)["out"]
        _, vl, va, sl, sa = hist[1]
        for name, m, got_l, got_a in (("val", masks[1], vl, va), ("test", masks[2], sl, sa)):
            cnt = int(m.sum())
            if not cnt:
                continue
            wl_, wa_ = nll(ev[m], y[m]).item(), int((ev[m].argmax(1) == y[m]).sum()) / cnt
            if abs(got_l - wl_) > 1e-4 * max(1.0, abs(wl_)) or abs(got_a - wa_) > 1.5 / cnt:
                bad.append((desc, f"{name} loss / accuracy of the last epoch: {(got_l, got_a)} vs {(wl_, wa_)}",
                            getattr(r, "engine", None) is not None))
        del r
    torch.save(bad, os.path.join(out_dir, f"runfuzz_{rank}.pt"))
    if own_group:
        leave_group()


def dist_fuzz_worker(rank, world, port, out_dir, prop_seeds, run_seeds):
    """propagate_fuzz_worker and runner_fuzz_worker in ONE set of rank processes (the interpreter start-up of the ranks is most
    of a short fuzz's time)."""
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("RGBX_TEST_DUMP_AFTER", "600")), exit=True)  # a hang ends with its stacks
    _init(rank, world, port)
    propagate_fuzz_worker(rank, world, port, out_dir, prop_seeds, own_group=False)
    runner_fuzz_worker(rank, world, port, out_dir, run_seeds, own_group=False)
    leave_group()
    faulthandler.cancel_dump_traceback_later()
