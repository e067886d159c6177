"""Data-parallel-over-nodes training of the reference's loop body (itexperiments.py:417-473) on a 1-D
node partition: one process per GPU, replicated parameters, per-propagate halo exchange, global
BatchNorm statistics, all-reduced loss and parameter gradients."""
import torch
import torch.nn.functional as F

from .comm import Comm
from .graph import install, install_replicas
from .nn import DistBatchNorm1d
from .plan import partition_bounds


class DistRunner:
    """model: any of the conv-stack models (GCN / GraphSAGE / GraphSAGE2 / APPNPStack), freshly built
    with the same seed on every rank. edge_index / x / y / masks: GLOBAL tensors (CPU or device); each
    rank keeps its node slice of x / y / masks and the whole edge list for index arithmetic."""

    def __init__(self, model, edge_index, x, y, masks, rank, world, device, lr=0.01, weight_decay=0.0,
                 comm=None, backend=None, exchange="auto", resident_features=True, pieces=None,
                 interleave_evals=True, fused=True, pieces_in=2, cache_input_aggregate=False, src_split=False,
                 pipeline=True, share_eval_forward=False):
        """`share_eval_forward`: the val and the test statistics of an epoch come from ONE eval forward (the reference's
        second eval forward, itexperiments.py:470, recomputes the outputs of its first, :464; experiment()'s default on one
        GPU since round 4) — a third of an epoch's forward exchanges less, the same five numbers."""
        self.comm = comm or Comm()
        self.rank, self.world, self.device = rank, world, device
        if "x" in str(exchange):  # an explicit R x C grid must factor the ranks: said before any structure is built
            R, C = (int(v) for v in str(exchange).split("x"))
            if R * C != world:
                raise ValueError(f"exchange={exchange!r} does not factor the world size {world}")
        N = x.size(0)
        lo, hi = partition_bounds(N, world)[rank:rank + 2]
        self.lo, self.hi, self.N = lo, hi, N
        self.x = x[lo:hi].to(device).contiguous()
        self.y = y[lo:hi].to(device)
        self.masks = [m[lo:hi].to(device) for m in masks]
        # the conv layers see (x_local, token): the token's cache entries are the DistGraphs. The global edge list is
        # on the device only until this rank's structures exist (release_edge_list)
        self.token = torch.zeros((2, 1), dtype=torch.int64, device=device)
        self.graphs = install(self.token, hi - lo, edge_index.to(device), N, self.comm, backend, exchange, pieces)
        import os
        self.interleave_evals = interleave_evals and world > 1 and os.environ.get("RGBX_INTERLEAVE", "auto") != "never"
        self.pipeline = bool(pipeline)  # epoch(more=True) may compute the next training step ahead (fused schedule)
        self.share_eval_forward = bool(share_eval_forward)
        self._spec = None
        self._streams = None
        self._epochs_done = 0
        self.host_enqueue_s = 0.0
        self._interleave_settled = False
        self.interleave_decision = None
        if device.type == "cuda":  # one timed all-to-all: the exchange cost model then uses this fabric's link rate
            self.link_gbs = self.comm.measure_link_gbs(device)
        # "replicate": every rank computes the first conv layer for all N nodes from the whole (static) feature
        # matrix and the second one for its own targets, with no activation exchange at all (graph.ReplicaGraph);
        # chosen by the cost model where the exchange would cost more than the redundant first layer
        self.replicated = self._choose_replicate(model, exchange, world)
        self.x_in = self.x  # what the model is called with
        if self.replicated:
            self.x_in = x.to(device).contiguous()
            self.replicas = install_replicas(self.token, self.graphs, N)
            for r in self.replicas.values():
                r.pin_resident(self.x_in)
        elif resident_features:
            for g in self.graphs.values():
                g.pin_resident(self.x)  # boundary rows of the static features are fetched once and kept
        self.model = DistBatchNorm1d.convert(model.to(device), self.comm)
        if self.replicated:
            for m in self.model.modules():
                if isinstance(m, DistBatchNorm1d):
                    m.replicated_rows = N
        # one fused multi-tensor launch per step on the GPU (the reference's torch.optim.Adam, itexperiments.py:391, same
        # update rule)
        self._fused_adam = device.type == "cuda"
        self._params = list(self.model.parameters())
        self.opt = torch.optim.Adam(self._params, lr=lr, weight_decay=weight_decay,
                                    **({"fused": True} if self._fused_adam else {}))
        for part, m in zip(("train", "val", "test"), self.masks):
            sel = self.y[m]
            if sel.numel() and int(sel.min()) < 0:  # NLLLoss on out[mask] raises on such rows in the reference
                raise RuntimeError(f"{part} mask selects nodes with a negative label (unlabelled)")
        cnt = torch.tensor([float(m.sum()) for m in self.masks], dtype=torch.float64, device=device)
        self.mask_counts = self.comm.all_reduce_sum_(cnt).tolist()
        # conv stacks under a column-slice exchange scheme run on the fused per-rank schedule (dist/stack.py: layer
        # outputs written straight into send buffers, BatchNorm / transform / loss in the return stage's kernel);
        # everything else — and `fused=False`, the conservative setting — goes through the modules
        self.engine = None
        if fused and not self.replicated and resident_features and world > 1:
            from .stack import GridStack
            self.engine = GridStack.build(self.model, self.graphs, self.comm, backend or self.graphs[0].backend, self.x,
                                          self.y, self.masks, self.mask_counts, pieces_in=pieces_in,
                                          cache_input_aggregate=cache_input_aggregate, src_split=src_split)
            # the schedule moves views (the list form of all-to-all): one small exchange with a known answer first — a
            # backend / build that does not deliver them as assumed must fail HERE, loudly, not train on wrong rows
            if self.engine is not None and self.interleave_evals:
                # under the interleaved schedule 2 outbound pieces beat the cost model's 4 at every link rate and latency
                # replayed: fewer, longer slice launches, and the pieces' exchanges are hidden by other
                # generators' work rather than by the next piece
                for g in self.graphs.values():
                    g.auto_pieces_cap = 2
            if self.engine is not None and not self.comm.self_test_views(device):
                raise RuntimeError("dist: the view all-to-all self-test failed on this backend (rows did not arrive "
                                   "where the fused schedule expects them); run with fused=False (bench.py --no-fused)")

    _REPLICABLE = {"GCNConv": 1, "SAGEConv": 0, "MySAGEConv": 2}  # conv class -> the loops mode its graph is keyed by

    def _choose_replicate(self, model, exchange, world):
        """True when the first two conv layers run under the replicate scheme: asked for (exchange="replicate") or,
        under "auto", modelled cheaper per epoch than the best exchange scheme (DistGraph.replicate_costs — the same
        inputs on every rank). Needs a conv stack whose first layer aggregates the input features before
        transforming them (in_channels <= out_channels), so that a layer's stage follows from its input."""
        convs = getattr(model, "convs", None)
        ok = (world > 1 and convs is not None and len(convs) >= 2
              and all(type(c).__name__ in self._REPLICABLE and getattr(c, "add_self_loops", True) for c in convs[:2])
              and convs[0].in_channels <= convs[0].out_channels)
        if exchange == "replicate":
            if not ok:
                raise RuntimeError("exchange='replicate' needs world > 1 and a GCN / GraphSAGE / GraphSAGE2 conv stack "
                                   "whose first layer has in_channels <= out_channels")
            return True
        if exchange != "auto" or not ok:
            return False
        g = self.graphs[self._REPLICABLE[type(convs[0]).__name__]]
        self.replicate_costs = g.replicate_costs(convs[0].in_channels, convs[0].out_channels)
        return self.replicate_costs["replicate"] < self.replicate_costs["exchange"]

    # ---- statistics used by bench.py --------------------------------------------------------
    def plan(self, loops_mode, kind):
        return self.graphs[loops_mode].plan(kind)

    def release_edge_list(self):
        """Call after the first epoch (every structure the model uses has been built by then): the global int64
        edge list leaves HBM; what stays is this rank's CSRs, exchange lists and resident boundary rows."""
        for g in self.graphs.values():
            g.release_edges()

    def _sync_grads(self):
        if self.engine is not None:  # the fused schedule wrote every gradient into one flat buffer (views as .grad)
            self.comm.all_reduce_sum_(self.engine.flat_grads)
            return
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.comm.all_reduce_sum_(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def _nll_sum(self, res, m):
        """Sum over this rank's masked rows of NLLLoss(log_softmax(emb)): from the logits on the GPU."""
        if res["emb"].is_cuda:
            from .. import ops
            return ops.masked_ce_loss(res["emb"], self.y, m, reduction="sum")
        return F.nll_loss(res["out"][m], self.y[m], reduction="sum")

    def train_step(self, sync=True):
        """One training step. `sync=False`: returns this rank's share of the loss as a device tensor [1]
        (float64) instead of the all-reduced Python float — epoch() reduces everything once."""
        self.discard_speculation()
        part = self._forward_backward()
        self._optimizer_step()
        if not sync:
            return part
        return self.comm.all_reduce_sum_(part.clone()).item()

    def _forward_backward(self):
        """Training forward + backward: every parameter's .grad holds this rank's share afterwards; returns this rank's
        share of the loss (float64 device tensor [1]). The optimizer step is the caller's (_optimizer_step)."""
        self.model.train()
        if self.engine is not None:
            return self.engine.train_step()  # forward + backward, every parameter's .grad (over)written in place
        self.opt.zero_grad()
        res = self.model(self.x_in, self.token)
        m = self.masks[0]
        loss = self._nll_sum(res, m) / self.mask_counts[0]
        loss.backward()
        return loss.detach().double().reshape(1)

    def _optimizer_step(self, grads_reduced=False):
        if not grads_reduced:
            self._sync_grads()
        self.opt.step()
        if self.engine is not None:
            self.engine.note_optimizer_step()
        # (a fused step writes the parameters without moving their version counters: ops.note_weights_changed, the
        # global optimizer post-hook, retires what is cached per parameter state)

    def discard_speculation(self):
        """Drop the next epoch's training step if epoch(more=True) has computed one ahead (a loop that stops early,
        a caller that steps by hand): it has written nothing but the gradient buffer."""
        if self._spec is not None:
            self._spec = None
            self.engine.discard_speculation()

    def evaluate(self, which, sync=True):
        """Eval forward + (masked NLL sum, correct count) of this rank's rows. `sync=True`: all-reduced and
        normalised Python floats (loss, accuracy, outputs); `sync=False`: the raw device tensor [2] and outputs."""
        self.model.eval()
        if self.engine is not None and not sync:
            return self.engine.eval_stats(which), None  # loss statistics straight from the last layer's kernel
        with torch.no_grad():
            res = self.model(self.x_in, self.token)
        stats = self._stats_of(res, self.masks[which])
        if not sync:
            return stats, res
        stats = (self.comm.all_reduce_sum_(stats) / self.mask_counts[which]).tolist()
        return stats[0], stats[1], res

    def _stats_of(self, res, m):
        """[masked NLL sum, correct count] (float64) of this rank's rows of one eval forward's outputs."""
        if res["emb"].is_cuda:
            from .. import ops
            return ops.masked_ce_accuracy(res["emb"], self.y, m)[::2]  # a view: a list index would go through the host
        out = res["out"]  # gloo / CPU tests
        return torch.stack([F.nll_loss(out[m], self.y[m], reduction="sum"),
                            (out[m].max(dim=1)[1] == self.y[m]).sum().float()]).double()

    def evaluate_pair(self):
        """(val statistics, test statistics) of this rank's rows from ONE eval forward (share_eval_forward): the fused
        schedule takes both masks in its last launch (GridStack.eval_shared); the module route forwards once and reads
        the logits under both masks."""
        self.model.eval()
        if self.engine is not None:
            return self.engine.eval_shared(1, 2)
        with torch.no_grad():
            res = self.model(self.x_in, self.token)
        return self._stats_of(res, self.masks[1]), self._stats_of(res, self.masks[2])

    def _interleaved_evals(self):
        """The val and the test forward (itexperiments.py:464-473: two full eval forwards per epoch, both run here)
        issued by two host threads that take turns at their exchange waits (comm.TakeTurns), each on its own HIP
        stream: one forward's all-to-all is in flight while the other forward aggregates. Same kernels, same
        numbers as two forwards in a row."""
        import threading
        from .comm import TakeTurns
        cuda = self.device.type == "cuda"
        # one side stream per forward where an exchange can be in flight beside kernels (RCCL; the emulated rank). The
        # gloo rehearsal stages every exchange through the host and blocks in it: nothing overlaps there, and with several
        # ranks SHARING one GPU the extra hardware queues (3 per rank, 4 ranks + the test's parent process) were seen to
        # leave two ranks' side streams unscheduled for minutes while their peers sat in the collective
        # (tests/test_gpu_dist.py, APPNP reshard(4) at workload S inside the suite; never alone) — there both forwards
        # enqueue on the caller's stream, still taking turns at their exchanges
        side = cuda and getattr(self.comm, "backend", "nccl") != "gloo"
        if side and self._streams is None:
            self._streams = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)]
        turns = TakeTurns(2)
        out, err = [None, None], [None, None]
        main = torch.cuda.current_stream(self.device) if cuda else None
        self.model.eval()
        # what both forwards share is made HERE, on the main stream, before either eval stream starts (each waits for
        # the main stream): the eval operands with the BatchNorms folded in are cached per model state, and the thread
        # that found them cached would otherwise read tensors another stream is still writing
        if self.engine is not None:
            self.engine._eval_weights()
        elif cuda and hasattr(self.model, "_eval_operands"):
            with torch.no_grad():
                self.model._eval_operands()

        def run(i):
            turns.enter(i)
            try:
                if cuda:
                    torch.cuda.set_device(self.device)
                if side:
                    self._streams[i].wait_stream(main)
                    with torch.cuda.stream(self._streams[i]):
                        out[i] = self.evaluate(1 + i, sync=False)[0]
                elif cuda:
                    with torch.cuda.stream(main):  # a new thread starts on the default stream: stay on the caller's
                        out[i] = self.evaluate(1 + i, sync=False)[0]
                else:
                    out[i] = self.evaluate(1 + i, sync=False)[0]
            except BaseException as exc:
                # This rank's other forward and every peer would wait for ever in the next all-to-all for the
                # collectives this forward no longer issues: in a real multi-rank job the error ends the PROCESS at once
                # (Comm.abort: traceback, exit status 70; the launcher then ends the peers), it is not carried to a join
                # that might never return. One-process runs (emulated rank, world 1) re-raise it in the caller.
                err[i] = exc
                self.comm.abort(exc, f"eval forward {i} of the interleaved pair")
            finally:
                turns.leave()

        self.model.eval()
        self.comm.turns = turns
        try:
            threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            self.comm.turns = None
        for e in err:
            if e is not None:
                raise e
        if side:
            for s in self._streams:
                main.wait_stream(s)
        return out

    def epoch(self, more=False):
        """1 train forward+backward+Adam, then val and test forwards, as the reference loop body. The five
        numbers the reference reads with .item() along the way are only used after the epoch: they are reduced
        over the ranks in ONE all-reduce and read back in ONE copy, so the queues drain once per epoch, not three
        times.
        `more` (fused schedule with interleaved evals): the caller will ask for another epoch unless this one's numbers
        stop the loop. The eval forwards of THIS epoch are then interleaved with the forward + backward of the NEXT
        epoch's training step (GridStack.eval_pair_and_next_step: all three read the parameters this epoch's optimizer
        step left), whose optimizer step is taken at the top of the next call — or never (discard_speculation). Same
        arithmetic per step, same five numbers; the exchanges of the training step travel while the eval forwards
        aggregate and the other way round."""
        import time
        t0 = time.perf_counter()
        enq0 = self.host_enqueue_s
        if self._spec is not None:  # the step computed ahead during the previous epoch's eval forwards
            tl, self._spec = self._spec, None
            self.engine.accept_speculation()
            self._optimizer_step(grads_reduced=True)  # (reduced at the end of the call that computed them, see below)
        else:
            tl = self.train_step(sync=False)
        # the first epoch builds what the eval forwards use lazily (cost tables, plans / CSRs of widths only the
        # no_grad path aggregates at) — on the MAIN stream, one forward after the other, so that no structure is
        # produced on one of the two eval streams and consumed on the other; interleaving starts with the second epoch
        if self.share_eval_forward and not (self.interleave_evals and self.engine is not None and more and self.pipeline):
            v, s = self.evaluate_pair()
        elif self.interleave_evals and self.engine is not None and more and self.pipeline:
            self.model.eval()
            step = self.engine.eval_shared_and_next_step if self.share_eval_forward else self.engine.eval_pair_and_next_step
            v, s, self._spec = step(1, 2)
            # the gradient all-reduce of the step computed ahead goes out NOW, behind this epoch's work, instead of at the
            # top of the next call, where the queue is empty and every launch is paid at the host's pace (it touches the
            # gradient buffer only: a dropped step drops it too)
            self._sync_grads()
        elif self.interleave_evals and self.engine is not None:
            # fused schedule: the two forwards interleaved on ONE thread and stream (GridStack.eval_pair)
            self.model.eval()
            v, s = self.engine.eval_pair(1, 2)
        elif self.interleave_evals and self._epochs_done > 0:
            v, s = self._interleaved_evals()
        else:
            v, _ = self.evaluate(1, sync=False)
            s, _ = self.evaluate(2, sync=False)
        packed = self.comm.all_reduce_sum_(torch.cat([tl, v, s]))
        # host time to ENQUEUE the epoch (before the one read-back that waits for the GPU): when this approaches the
        # epoch's wall time the rank is host-bound, not kernel-bound (bench.py reports it per rank)
        self.host_enqueue_s += time.perf_counter() - t0
        p = packed.tolist()
        self._epochs_done += 1
        self._settle_interleave(self.host_enqueue_s - enq0, time.perf_counter() - t0)
        cv, cs = self.mask_counts[1], self.mask_counts[2]
        # (an eval mask without a row: nan, as the mean over an empty selection is on one GPU — not a ZeroDivisionError)
        nan = float("nan")
        return (p[0], p[1] / cv if cv else nan, p[2] / cv if cv else nan, p[3] / cs if cs else nan, p[4] / cs if cs else nan)

    def _settle_interleave(self, enqueue_s, wall_s):
        """Two host threads issuing the eval forwards hide one forward's exchange behind the other's aggregation, but
        cost host time (3.7 instead of 1.5 ms per epoch for rank 0 of 8 on the benchmark). On a slow or
        crowded host that makes the rank HOST-bound: when, in the second interleaved epoch, enqueueing took more than
        80 % of the epoch's wall time on ANY rank (one small all-reduce: every rank must take the same decision), the
        evals run one after the other from then on. RGBX_INTERLEAVE=always | never overrides."""
        if (not self.interleave_evals or self._interleave_settled or self._epochs_done != 3
                or self.engine is not None):  # the fused schedule interleaves without a second thread
            return
        self._interleave_settled = True
        import os
        mode = os.environ.get("RGBX_INTERLEAVE", "auto")
        if mode == "always":
            return
        ratio = torch.tensor([1.0 if mode == "never" else enqueue_s / max(wall_s, 1e-9)], dtype=torch.float64,
                             device=self.device)
        worst = self.comm.all_reduce_max_(ratio).item()
        self.interleave_decision = {"host_enqueue_over_wall": worst, "kept": worst <= 0.8}
        if worst > 0.8:
            self.interleave_evals = False

    def logits(self, training=False):
        self.model.train(training)
        with torch.no_grad():
            return self.model(self.x_in, self.token)["emb"]
