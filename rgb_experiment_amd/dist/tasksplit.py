"""The epoch split by TASK over two groups of ranks (new capability: the reference is single-device,
itexperiments.py:246): ranks [0, P/2) run the training step, ranks [P/2, P) the val and the test forward of the
reference loop body (itexperiments.py:427-440 and :464-473), each group holding the WHOLE graph node-partitioned over its
P/2 ranks (a DistRunner of its own on a sub-communicator).

Why: a column-slice scheme cuts every feature row into P slices, and below 32 floats a slice costs the same 128-byte line
per gathered row as a 32-float one — at P = 8 and d = 128 an APPNP rank (K = 10: all ten steps between one
pair of transposes, on width-16 slices of ALL rows) does twice the work of an ideal share. The reference epoch is four
independent K-step chains (training forward, its backward, val forward, test forward); with two groups of 4 ranks each
chain runs at width 32, and a rank runs two chains instead of four.

What makes the groups run at the same time: the eval forwards of epoch t and the forward + backward of epoch t + 1's
training step read the same parameters (what the optimizer step of epoch t left). So after its optimizer step the
training group hands the parameters and BatchNorm buffers to the eval group (one small broadcast) and, if the caller
announces another epoch, computes the next step's gradients AHEAD while the eval group evaluates — speculatively: the
model of the training ranks is exactly what the finished epoch left (the step's BatchNorm running statistics are set
aside, its gradients sit in .grad) until the next call accepts it (optimizer step) or discard_speculation drops it.
Same five numbers per epoch as DistRunner.epoch, same arithmetic per step."""
import torch
import torch.distributed as dist

from .. import ops
from .comm import Comm, EmulatedComm
from .runner import DistRunner


def pays(model, world, width=None):
    """Task split is worth it where halving the group doubles the slice width below the 128-byte line: an APPNP stack
    (its K steps dominate the epoch and run on width / P slices of the propagated matrix, width = output_dim:
    reference models/appnp_stack.py:29) on an even number of at least 4 ranks."""
    if world == 2:
        # two ranks share ONE xGMI link: any exchange scheme moves more than 250 MB per propagate over it (34 ms per epoch
        # at the benchmark's size) and the exchange-free replicate scheme repeats the first layer on every rank
        # (0.79 of a single-GPU epoch). One rank training and one evaluating, each on the whole graph with
        # the single-GPU kernels and no exchange at all, is max(training step, two eval forwards) = 0.5 of it.
        return True
    if type(model).__name__ != "APPNPStack" or world < 4 or world % 2:
        return False
    width = model.lin2.out_features if width is None else width
    return width // world < 32 and width // (world // 2) >= 16


def whole_graph_bytes(n, e, widths):
    """HBM one rank needs to hold the WHOLE graph and run the single-GPU kernels on it (the world == 2 task split):
    int64 edge_index, forward + transposed CSR (col, perm, weight per slot, rowptr), and per layer width the
    activations an epoch keeps (input, aggregate, output, their gradients: 6 matrices of [n, width] fp32, measured
    9.7 GB peak at n = 2 M, e = 60 M, widths 128: this estimate gives 9.9)."""
    slots = e + n
    return 16 * e + 2 * (12 * slots + 4 * (n + 1)) + sum(6 * 4 * n * w for w in widths)


def whole_graph_fits(n, e, widths, device, comm=None):
    """Every rank must take the same decision: the smallest free HBM over the ranks decides (one small all-reduce)."""
    if device.type != "cuda":
        return True
    free = torch.tensor([float(torch.cuda.mem_get_info(device)[0])], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.all_reduce(free, op=dist.ReduceOp.MIN)
        else:
            h = free.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN)
            free = h
    return whole_graph_bytes(n, e, widths) <= 0.8 * float(free.item())


def resolve(mode, model, world, n, e, widths, device):
    """task_split = 'auto' | 'on' | 'off' (experiment(task_split=...), bench.py --task-split, environment
    RGBX_TASK_SPLIT overrides 'auto'). auto: where it pays (`pays`), except that on TWO ranks — where the split means
    the whole graph on both GPUs, i.e. no memory scaling — it is taken only when one GPU can hold the whole graph;
    otherwise the node-partitioned DistRunner runs (each rank holds half of everything)."""
    import os
    if mode in (None, "auto"):
        mode = os.environ.get("RGBX_TASK_SPLIT", "auto")
    if mode not in ("auto", "on", "off"):
        raise ValueError(f"task_split must be 'auto', 'on' or 'off', got {mode!r}")
    if mode == "off" or world < 2 or world % 2:
        return False
    if mode == "on":
        return True
    if not pays(model, world):
        return False
    return world != 2 or whole_graph_fits(n, e, widths, device)


class WholeGraphRunner:
    """What TaskSplitRunner needs of a group of ONE rank: the reference's training step and eval forwards on the whole
    graph, on this rank's GPU, with the single-GPU kernels (fused aggregate + transform, loss inside the last layer's
    kernel: models/_stack.masked_ce) — no partition, no exchange."""

    engine, replicated, interleave_evals, interleave_decision, graphs = None, False, False, None, None

    def __init__(self, model, edge_index, x, y, masks, device, lr=0.01, weight_decay=0.0, share_eval_forward=False):
        self.device = device
        self.share_eval_forward = bool(share_eval_forward)
        self.model = model.to(device)
        self.fwd = {"x": x.to(device).contiguous(), "edge_index": edge_index.to(device)}
        self.y = y.to(device)
        self.masks = [m.to(device) for m in masks]
        self.mask_counts = [float(m.sum()) for m in self.masks]
        self.N = x.size(0)
        self.lo, self.hi = 0, self.N
        self._params = list(self.model.parameters())
        # torch.optim.Adam exactly as experiment() builds it on one GPU (itexperiments.py: capturable; reference :391) — not the
        # fused flavour the partitioned runners take: the training rank then produces the SAME BITS as a one-GPU run
        # (same kernels, same optimizer arithmetic), and two GPUs return the model one GPU returns
        self._fused_adam = False
        self.opt = torch.optim.Adam(self._params, lr=lr, weight_decay=weight_decay, capturable=device.type == "cuda")
        self._epochs_done = 0
        self.host_enqueue_s = 0.0

    def _forward_backward(self):
        from ..models._stack import masked_ce
        self.model.train()
        self.opt.zero_grad()
        loss = masked_ce(self.model, self.fwd, self.y, self.masks[0])[0]
        loss.backward()
        return loss.detach().double().reshape(1)

    def _optimizer_step(self):
        self.opt.step()  # ops.note_weights_changed (global optimizer post-hook) retires the cached W^T

    def evaluate(self, which, sync=False):
        from ..models._stack import masked_ce
        self.model.eval()
        with torch.no_grad():
            stats = masked_ce(self.model, self.fwd, self.y, self.masks[which])[1]  # [nll sum, rows, correct]
        return stats[::2], None

    def evaluate_pair(self):
        """(val statistics, test statistics) from ONE eval forward (models/_stack.masked_ce_pair)."""
        from ..models._stack import masked_ce_pair
        self.model.eval()
        with torch.no_grad():
            st = masked_ce_pair(self.model, self.fwd, self.y, self.masks[1], self.masks[2])
        return st[0, ::2], st[1, ::2]

    def logits(self, training=False):
        self.model.train(training)
        with torch.no_grad():
            return self.model(**self.fwd)["emb"]

    def release_edge_list(self):
        pass


class TaskSplitRunner:
    """See the module docstring. `epoch(more=...)` / `discard_speculation()` / `logits()` as DistRunner."""

    def __init__(self, model, edge_index, x, y, masks, rank, world, device, lr=0.01, weight_decay=0.0, comm=None,
                 role=None, **runner_kw):
        if world < 2 or world % 2:
            raise RuntimeError(f"task split needs an even number of ranks, got {world}")
        half = world // 2
        ex = str(runner_kw.get("exchange", "auto"))
        if "x" in ex:
            # refused HERE, by every rank at once: left to the first propagate, the training group would raise while the eval
            # group already waits in the hand-over broadcast (found by the gloo fuzz, round 4: "2x2" on 4 ranks = groups of 2)
            R, C = (int(v) for v in ex.split("x"))
            if R * C != half:
                raise ValueError(f"exchange={ex!r} does not factor the {half} ranks of a task-split group "
                                 f"({world} ranks = 2 groups of {half})")
        self.rank, self.world_size, self.device = rank, world, device
        self.role = role or ("train" if rank < half else "eval")
        if isinstance(comm, EmulatedComm):  # bench.py --emulate-rank: one process stands for one rank of one group
            self.world = comm
            inner = EmulatedComm(half)
        else:
            self.world = comm or Comm()
            # every rank creates BOTH groups, in the same order (torch.distributed's rule)
            groups = [dist.new_group(list(range(half))), dist.new_group(list(range(half, world)))]
            inner = Comm(groups[0 if self.role == "train" else 1])
        if half == 1 and device.type == "cuda" and runner_kw.get("backend") is None:
            self.inner = WholeGraphRunner(model, edge_index, x, y, masks, device, lr=lr, weight_decay=weight_decay,
                                          share_eval_forward=runner_kw.get("share_eval_forward", False))
        else:  # (a group of one rank on the CPU: the gloo rehearsal, through the partitioned path with its injected aggregator)
            self.inner = DistRunner(model, edge_index, x, y, masks, rank % half, half, device, lr=lr,
                                    weight_decay=weight_decay, comm=inner, pipeline=False, **runner_kw)
        self.model = self.inner.model
        self.lo, self.hi, self.N = self.inner.lo, self.inner.hi, self.inner.N
        self.masks, self.y, self.mask_counts = self.inner.masks, self.inner.y, self.inner.mask_counts
        self.engine = None
        # the tensors themselves, not their .data aliases: an alias has a version counter of its own, and what is cached
        # per parameter version (folded eval operands, transposed weights) must see the eval ranks' copies arrive
        self._state = list(self.model.parameters()) + list(self.model.buffers())
        self._bns = [m for m in self.model.modules() if isinstance(m, torch.nn.BatchNorm1d)]
        self._spec = None          # loss share of a step computed ahead
        self._spec_bn = None       # its BatchNorm buffers, set aside until the step is accepted
        self.host_enqueue_s = 0.0

    # ---- hand-over of the model state -----------------------------------------------------------------------------
    def _hand_over(self):
        """Parameters and buffers of the training group's model -> every rank (source: rank 0; the training ranks hold
        identical copies, the eval ranks install what arrives)."""
        if isinstance(self.world, EmulatedComm):
            return
        # one flat buffer per dtype present: every tensor travels in ITS OWN dtype (a float64 or bfloat16 buffer of a
        # user's module is not squeezed through float32)
        for dtype in sorted({t.dtype for t in self._state}, key=str):
            ts = [t for t in self._state if t.dtype == dtype]
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            self._broadcast(flat)
            if self.role == "eval":
                off = 0
                with torch.no_grad():
                    for t in ts:
                        t.copy_(flat[off:off + t.numel()].view_as(t))  # copy_ moves the version counters: caches follow
                        off += t.numel()

    def _broadcast(self, flat):
        if self.world.backend == "nccl" or not flat.is_cuda:
            dist.broadcast(flat, 0, group=self.world.group)
            return
        h = flat.cpu()
        dist.broadcast(h, 0, group=self.world.group)
        flat.copy_(h)

    # ---- the speculative step of the training group ------------------------------------------------------------------
    def _bn_buffers(self):
        return [(bn.running_mean, bn.running_var, bn.num_batches_tracked) for bn in self._bns
                if bn.running_mean is not None]

    def _step_ahead(self):
        """Forward + backward of the NEXT training step: gradients stay in .grad, the running statistics the forward
        moved are set aside and the model's own are put back."""
        r = self.inner
        before = [tuple(t.clone() for t in trio) for trio in self._bn_buffers()]
        part = r._forward_backward()
        after = []
        for trio, old in zip(self._bn_buffers(), before):
            after.append(tuple(t.clone() for t in trio))
            for t, o in zip(trio, old):
                t.copy_(o)
        self._spec, self._spec_bn = part, after

    def discard_speculation(self):
        self._spec = self._spec_bn = None

    # ---- one epoch --------------------------------------------------------------------------------------------------
    def epoch(self, more=False):
        import time
        t0 = time.perf_counter()
        r = self.inner
        dev = self.device
        zeros = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
        if self.role == "train":
            if self._spec is not None:  # the step computed during the previous epoch's eval forwards
                tl, self._spec = self._spec, None
                for trio, new in zip(self._bn_buffers(), self._spec_bn):
                    for t, n in zip(trio, new):
                        t.copy_(n)
                self._spec_bn = None
            else:
                tl = r._forward_backward()
            r._optimizer_step()
            self._hand_over()
            if more:
                self._step_ahead()
            v, s = zeros(2), zeros(2)
        else:
            self._hand_over()
            ops.note_weights_changed()  # parameters written from outside (broadcast into raw storage)
            r.model.eval()
            e0 = time.perf_counter()
            if getattr(r, "share_eval_forward", False):
                v, s = r.evaluate_pair()  # one eval forward, both masks
            elif getattr(r, "engine", None) is not None and r._epochs_done > 0:
                # a fused schedule inside the group (--task-split on for a conv stack): its ONE-thread interleave of the
                # two forwards; the two-thread path below would run over engine state that is not thread-safe
                v, s = r.engine.eval_pair(1, 2)
            elif r.interleave_evals and r._epochs_done > 0:
                v, s = r._interleaved_evals()
            else:
                v, _ = r.evaluate(1, sync=False)
                s, _ = r.evaluate(2, sync=False)
            r._epochs_done += 1
            tl = zeros(1)
            eval_enqueue_s = time.perf_counter() - e0
        # the five numbers of the epoch: the loss shares come from the training ranks, the eval statistics from the eval
        # ranks (everybody else adds zeros): ONE all-reduce over all ranks, one read-back
        packed = self.world.all_reduce_sum_(torch.cat([tl.reshape(1).double(), v.double(), s.double()]))
        self.host_enqueue_s += time.perf_counter() - t0
        p = packed.tolist()
        if self.role == "eval" and hasattr(r, "_settle_interleave"):
            # the host-bound fallback of DistRunner.epoch (two eval threads cost host time): decided once, over the eval
            # group's communicator, from the time it took to enqueue the two forwards against the epoch's wall time
            r._settle_interleave(eval_enqueue_s, time.perf_counter() - t0)
        cv, cs = self.mask_counts[1], self.mask_counts[2]
        # (an eval mask without a row: nan, as the mean over an empty selection is on one GPU — not a ZeroDivisionError)
        nan = float("nan")
        return (p[0], p[1] / cv if cv else nan, p[2] / cv if cv else nan, p[3] / cs if cs else nan, p[4] / cs if cs else nan)

    # what bench.py reads of a runner
    graphs = property(lambda self: self.inner.graphs)
    replicated = property(lambda self: self.inner.replicated)
    replicas = property(lambda self: getattr(self.inner, "replicas", None))
    interleave_evals = property(lambda self: self.inner.interleave_evals)
    interleave_decision = property(lambda self: self.inner.interleave_decision)
    link_gbs = property(lambda self: getattr(self.inner, "link_gbs", None))

    def logits(self, training=False):
        """This rank's rows (of its GROUP's partition) of the model's logits."""
        return self.inner.logits(training)

    def release_edge_list(self):
        self.inner.release_edge_list()
