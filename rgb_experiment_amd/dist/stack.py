"""Conv-stack models (GCN / GraphSAGE / GraphSAGE2: reference models/gcn.py:25-31, graphsage.py:26-32,
graphsage2.py:27-33) on a node partition, with the single-GPU fusions carried over to the row-group x column-slice
exchange (DistGraph "gridRxC" / "reshard"). New capability — the reference is single-device (itexperiments.py:246);
the arithmetic of every layer is the one of rgb_experiment_amd.nn.conv / ops, only WHERE each piece runs differs.

What a rank runs per training step of an L-layer stack (layer 0 on the resident input features, layers >= 1
exchanged; no pack / unpack pass anywhere):

  layer 0   ONE rgbx_fused_layer_f32 launch over the [local; halo] CSR: aggregate + transform (+ root term), the raw
            output h_0 written row-major (BatchNorm's backward reads it) AND blocked — column slice by column slice —
            straight into the send buffer of the next layer's exchange; the aggregate z_0 stored for dW_0; the column
            sums of h_0 (BatchNorm statistics) from the MFMA tiles. The [2, d] sums are all-reduced, one finalize launch.
  layer i   inbound all-to-all of the RAW slices (BatchNorm is applied by the consumer, by linearity of the
            aggregation) -> one rgbx_spmm_csr_f32 launch per outbound piece over the row group's rows at slice
            width, each into its send buffer, its all-to-all in flight while the next piece is aggregated; the pieces
            land directly in a blocked [C, n_local, d/C] buffer ->
            ONE rgbx_fused_layer_f32 launch in DENSE mode (the return stage): loads the 32-row tiles from the
            received slices, applies z = s * u + t * rowsum(P) (BatchNorm of the layer's input), stores z_i (dW_i),
            multiplies by W_i on the MFMA units, adds the root term from h_{i-1}, and — last layer — turns the
            logits tile into the masked cross-entropy statistics and the loss gradient; middle layers write h_i
            row-major + blocked and take its column sums.
  backward  per exchanged layer: dW_i = dy^T z_i (+ column sums = db_i) on rgbx_gemm_tn_f32; q = dy W_i written blocked
            into the send buffer by a DENSE launch; exchange + transposed SpMM pieces + return; the received slices are
            turned into rows once (rgbx_blocked_to_rows_f32) for BatchNorm's backward kernels (all-reduce of [2, d]);
            layer 0: dW_0 = g^T z_0.
An eval forward folds every BatchNorm into the preceding layer's weights and writes no row-major activation at all
(layer outputs exist only as send buffers); the last launch returns the loss statistics, no logits.

Every exchange moves views (Comm.all_to_all_views: grouped send / recv of the slices where they lie), so the same
slice buffer serves the R ranks that need it without being duplicated.
"""
import torch

from ..nn import batchnorm as B
from .plan import partition_bounds


class _AllOf:
    """Several exchange works waited for as one."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def _drive(gen):
    """Run a schedule generator to its end: it yields every exchange it is about to depend on, the driver waits for it.
    (GridStack.eval_pair drives TWO such generators alternately instead, so that one forward's exchange is in flight
    while the other forward's kernels are enqueued.)"""
    try:
        work = next(gen)
        while True:
            work.wait()
            work = gen.send(None)
    except StopIteration as done:
        return done.value


class HipStackBackend:
    """Product compute backend of GridStack: rgbx_fused_layer_f32, rgbx_spmm_csr_f32, rgbx_gemm_tn_f32,
    rgbx_blocked_to_rows_f32 (through rgb_experiment_amd.ops). `agg` = the HipAggregator the DistGraph was built with."""

    def __init__(self, agg):
        self.agg = agg

    def layer(self, x, wt, handle=None, rows=None, **kw):
        """ops.fused_layer with `handle` = (csr, per-slot weights) of DistGraph.backend.prepare; `rows` = (lo, hi)
        restricts an aggregating launch to that row range of the CSR (the outputs then have hi - lo rows)."""
        from .. import graph as G
        from .. import ops
        csr = w = None
        if handle is not None:
            csr, w = handle
            if rows is not None:
                if csr.split is not None:
                    raise RuntimeError("row ranges of a CSR with a hub-row plan are not supported")
                lo, hi = rows
                csr = G.CSR(csr.rowptr[lo:hi + 1], csr.col, csr.perm, hi - lo, csr.nnz, None)
        return ops.fused_layer(x, wt, csr=csr, w=w, **kw)

    def run_rows(self, handle, x, lo, hi, out, kind, accumulate=False):
        return self.agg.run_rows(handle, x, lo, hi, out, kind=kind, accumulate=accumulate)

    def run(self, handle, x, kind):
        return self.agg.run(handle, x, kind=kind)

    def rows_ok(self, handle):
        """Row ranges of this CSR can be launched on their own (no hub-row plan: its row ids are absolute)."""
        return handle[0].split is None

    def gemm_tn(self, a, b, colsum=False, out=None, sums_out=None):
        from .. import ops
        return ops.gemm_tn(a, b, colsum=colsum, out=out, sums_out=sums_out)

    def blocked_to_rows(self, blk, bias=None):
        from .. import ops
        return ops.blocked_to_rows(blk, bias=bias)

    def ce_stats(self, logits, y, mask):
        """[nll sum, selected rows, hits] (float64, device) of materialised logits."""
        from .. import ops
        return ops.masked_ce_accuracy(logits, y, mask)

    def ce_stats_blocked(self, blk, bias, y, mask):
        """The same of logits = received slices + bias, read in place (selected rows only; C % 4 == 0, C <= 256: what
        GridStack.build admits for the last layer)."""
        from .. import ops
        return ops.masked_ce_accuracy_blocked(blk, y, mask, bias=bias)


class LayerSpec:
    """What GridStack needs of one conv layer: aggregation kind, rewrite mode, the parameters and how they enter
    out = (P x) W^T + sum(biases) (+ x Wr^T)."""

    def __init__(self, conv):
        from ..graph import LOOPS_ADD_REMAINING, LOOPS_KEEP, LOOPS_REMOVE_ADD
        name = type(conv).__name__
        self.d_in, self.d_out = conv.in_channels, conv.out_channels
        if name == "GCNConv":
            self.kind, self.mode = "gcn", LOOPS_ADD_REMAINING
            self.W, self.biases, self.Wr = conv.lin.weight, [conv.bias], None
        elif name == "SAGEConv":
            self.kind, self.mode = "mean", LOOPS_KEEP
            self.W, self.biases, self.Wr = conv.lin_l.weight, [conv.lin_l.bias], conv.lin_r.weight
        elif name == "MySAGEConv" and conv.add_self_loops:
            self.kind, self.mode = "mean", LOOPS_REMOVE_ADD
            self.W, self.biases, self.Wr = conv.lin_l.weight, [conv.lin_l.bias, conv.lin_r.bias], conv.lin_r.weight
        else:
            raise TypeError(name)

    def bias(self):
        b = self.biases[0].detach()
        for extra in self.biases[1:]:
            b = b + extra.detach()
        return b


def _supported(K, n_out, root):
    return K >= 4 and K % 4 == 0 and K <= 256 and n_out >= 32 and n_out % 32 == 0 and (not root or n_out <= 256)


class GridStack:
    """See the module docstring. Built by `GridStack.build` (None when the model / scheme / widths do not fit; the
    caller then runs the generic module path, which is what every other model uses)."""

    @classmethod
    def build(cls, model, graphs, comm, backend, x_local, y, masks, mask_counts, pieces_in=1,
              cache_input_aggregate=False, src_split=False):
        convs, bns = getattr(model, "convs", None), getattr(model, "bns", None)
        if comm.world < 2 or convs is None or bns is None or len(convs) < 2 or len(bns) != len(convs) - 1:
            return None
        try:
            specs = [LayerSpec(c) for c in convs]
        except TypeError:
            return None
        if len({s.mode for s in specs}) != 1 or len({s.kind for s in specs}) != 1:
            return None
        from .nn import DistBatchNorm1d
        if not all(isinstance(b, DistBatchNorm1d) and b.affine and b.track_running_stats for b in bns):
            return None
        stack_be = getattr(backend, "stack_backend", None)
        if stack_be is None:
            from .graph import HipAggregator
            if not (isinstance(backend, HipAggregator) and x_local.is_cuda):
                return None
            stack_be = HipStackBackend(backend)
        else:
            stack_be = stack_be()
        dg = graphs[specs[0].mode]
        first = specs[0]
        if not (dg.is_resident(x_local) and first.d_in <= first.d_out
                and _supported(first.d_in, first.d_out, first.Wr is not None)):
            return None
        shapes = []
        for i, s in enumerate(specs[1:], start=1):
            shape = dg.shape(s.d_in)
            if shape is None:  # halo scheme for this width
                return None
            dc = s.d_in // shape[1]
            if s.d_in % shape[1] or dc % 4 or not _supported(s.d_in, s.d_out, s.Wr is not None):
                return None
            if not _supported(s.d_out, s.d_in, False):  # q = dy W of the backward pass, on the same kernel
                return None
            shapes.append(shape)
        if specs[-1].d_out > 128:  # the loss epilogue's limit
            return None
        return cls(model, specs, bns, dg, shapes, comm, stack_be, x_local, y, masks, mask_counts, pieces_in,
                   cache_input_aggregate, src_split)

    def __init__(self, model, specs, bns, dg, shapes, comm, be, x_local, y, masks, mask_counts, pieces_in,
                 cache_input_aggregate=False, src_split=False):
        self.model, self.specs, self.bns, self.dg, self.comm, self.be = model, specs, list(bns), dg, comm, be
        self.shapes = [None] + shapes  # per layer: (R, C) of its exchange
        self.x, self.y, self.masks = x_local, y, masks
        self.n_loc, self.N, self.P = dg.n_local, dg.N_global, comm.world
        self.bounds = partition_bounds(self.N, self.P)
        dev = x_local.device
        self.grad_scale = torch.tensor([1.0 / mask_counts[0]], dtype=torch.float32, device=dev)
        self.mask_counts = mask_counts
        self.pieces_in = max(1, int(pieces_in))
        self.src_split = bool(src_split)
        # Eval forwards of a stack WITHOUT root terms (GCN): the last layer's transform is applied by the layer before it
        # (the eval-mode model is linear between the aggregations: P((z W'^T + b') W_last^T) = P(z (W_last W')^T + W_last b')),
        # so the return stage of the last exchange is bias + loss statistics with no product left — on a rank whose gathers
        # run elsewhere that product is exposed fp32 MFMA time. Same numbers to rounding
        # (reassociation, as the BatchNorm fold). Needs the slices of the logits to be exchangeable: width % C, % 4.
        last, prev = specs[-1], specs[-2]
        C_last = shapes[-1][1]
        self.last_transform_first = (all(sp.Wr is None for sp in specs) and last.d_out % C_last == 0
                                     and (last.d_out // C_last) % 4 == 0 and _supported(prev.d_in, last.d_out, False))
        self._rowsum = None
        # OPT-IN (never the headline): keep this rank's rows of P x, the first layer's aggregate of the static input
        # features — the same matrix in every forward of every epoch; layer 0 is then a DENSE launch over it
        self.cache_input_aggregate = bool(cache_input_aggregate)
        self._z0 = None
        self._folded = (None, None)
        self._steps = 0
        self._pair_tag = ""
        self._pending_bn, self._commit_list, self._opt_steps = [], [], 0
        self._pieces_first = None
        # every parameter gradient is a view of ONE flat buffer, written in place by the kernels that produce it: the
        # gradient all-reduce then takes the buffer as it is (no flatten / copy-back launches), and nothing is zeroed
        # between steps because every entry is overwritten
        params = [p for p in model.parameters()]
        self.flat_grads = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        off = 0
        self._grad_views = {}
        for p in params:
            self._grad_views[id(p)] = self.flat_grads[off:off + p.numel()].view_as(p)
            off += p.numel()

    def _g(self, p):
        """The gradient view of parameter `p` (installed as p.grad)."""
        v = self._grad_views[id(p)]
        if p.grad is not v:
            p.grad = v
        return v

    # ---- structures ------------------------------------------------------------------------------------------
    def _first(self):
        """([local; halo] CSR handle, the resident extended feature matrix) of layer 0."""
        kind = self.specs[0].kind
        handle, half = self.dg._ext_csr(kind)
        return handle, self.dg._resident_ext(self.x, half, kind)

    def rowsum(self):
        """Per own target row the sum of its aggregation weights (what a constant column contributes to the
        aggregate): the consumer needs it to push BatchNorm's shift through the aggregation."""
        if self._rowsum is None:
            handle, x_ext = self._first()
            ones = torch.ones((x_ext.size(0), 1), dtype=torch.float32, device=x_ext.device)
            self._rowsum = self.be.run(handle, ones, kind="rowsum").reshape(-1).contiguous()
        return self._rowsum

    def _blocked_buffer(self, i, rows=None, width=None):
        """Send / receive buffer of layer i's exchange: [C, rows, width / C] (width: the layer's input width unless the
        exchanged matrix is something else, see _eval_weights)."""
        C = self.shapes[i][1]
        width = self.specs[i].d_in if width is None else width
        return torch.empty((C, self.n_loc if rows is None else rows, width // C), dtype=torch.float32,
                           device=self.x.device)

    # ---- the exchange ----------------------------------------------------------------------------------------
    def _inbound(self, i, blk, piece=None, cols=None, behind_producer=False):
        """Column slice c of every node's row -> `cols` [N, dc] (c = this rank's slice): every peer q = (r', c') is sent
        slice c' of my rows — the same view for the R ranks that share it. `piece` = (k, n): only the k-th of n row
        pieces of every rank's rows (rows [n_q k / n, n_q (k + 1) / n) of rank q), so that the exchange of a piece can
        start as soon as its producer has written it. Returns (cols, work)."""
        R, C = self.shapes[i]
        P, b = self.P, self.bounds
        dc = blk.size(2)
        if cols is None:
            cols = torch.empty((self.N, dc), dtype=torch.float32, device=blk.device)
        k, n = piece or (0, 1)
        cut = lambda rows, j: rows * j // n
        a0, a1 = cut(self.n_loc, k), cut(self.n_loc, k + 1)
        send = [blk[q % C, a0:a1] for q in range(P)]
        recv = [cols[b[q] + cut(b[q + 1] - b[q], k):b[q] + cut(b[q + 1] - b[q], k + 1)] for q in range(P)]
        tag = f"in {k + 1}/{n}" + (" producer" if behind_producer else "") + self._pair_tag
        return cols, self.comm.all_to_all_views(send, recv, tag=tag)

    def _first_layer_pieces(self, handle):
        """Row pieces layer 0 is launched in — the SAME number on every rank (each piece is followed by an exchange): a
        rank whose [local; halo] CSR has a hub-row plan cannot launch row ranges of it, and then nobody does. Agreed
        once, by one small all-reduce at the first layer-0 launch (every rank makes it at the same point)."""
        if self._pieces_first is None:
            mine = torch.tensor([0.0 if self.be.rows_ok(handle) else 1.0], dtype=torch.float64, device=self.x.device)
            anyone = self.comm.all_reduce_max_(mine).item() > 0
            self._pieces_first = 1 if anyone else self.pieces_in
        return self._pieces_first

    def _first_layer(self, wt, bias, wtr, train):
        """Layer 0 on the resident features, launched in `pieces_in` row pieces; each piece's slices leave for their
        consumers as soon as its launch is enqueued (the all-to-all runs on RCCL's stream while the next piece is
        computed), so only the last piece's share of the inbound exchange is exposed. Training also keeps the raw
        rows (BatchNorm's backward), the aggregate (dW_0) and the column sums (BatchNorm's statistics).
        Returns (blk, (cols, works), h, z, colsums)."""
        sp = self.specs[0]
        handle, x_ext = self._first()
        n, dev = self.n_loc, self.x.device
        blk = self._blocked_buffer(1, width=wt.size(1))
        h = torch.empty((n, sp.d_out), dtype=torch.float32, device=dev) if train else None
        cached = self.cache_input_aggregate
        if cached and self._z0 is None:
            self._z0 = self.be.run(handle, x_ext, kind=f"{sp.kind}_fwd")
        z = self._z0 if cached else (torch.empty((n, sp.d_in), dtype=torch.float32, device=dev) if train else None)
        pieces = self._first_layer_pieces(handle)
        cols, works, cs = None, [], None
        for k in range(pieces):
            a, b = n * k // pieces, n * (k + 1) // pieces
            rows = None if pieces == 1 else (a, b)
            if cached:  # transform of the kept aggregate: no gather at all
                _, _, c = self.be.layer(z[a:b], wt, bias=bias, x_root=self.x[a:b] if wtr is not None else None,
                                        wt_root=wtr, want_out=False, out=None if h is None else h[a:b],
                                        want_colsums=train, out_blocked=blk[:, a:b],
                                        kind="cached_aggregate_linear_fwd")
            else:
                _, _, c = self.be.layer(x_ext, wt, handle=handle, rows=rows, bias=bias,
                                        x_root=self.x[a:b] if wtr is not None else None, wt_root=wtr, want_out=False,
                                        out=None if h is None else h[a:b], z=None if z is None else z[a:b],
                                        want_colsums=train, out_blocked=blk[:, a:b], kind=f"{sp.kind}_linear_fwd")
            cs = c if cs is None or c is None else cs + c
            cols, work = self._inbound(1, blk, piece=(k, pieces), cols=cols, behind_producer=True)
            works.append(work)
        return blk, (cols, works), h, z, cs

    def _propagate(self, i, direction, blk, inbound=None):
        return _drive(self._propagate_g(i, direction, blk, inbound))

    def _propagate_g(self, i, direction, blk, inbound=None, landed=False):
        """The exchanged aggregation of layer i: blocked rows of this rank in, blocked aggregated rows of this rank
        out (u[c', r, :] = column slice c' of own row r of P x, or of P^T x for direction "bwd"). `inbound` = (cols,
        [work per source piece]) when the caller has already issued the inbound exchange (piece by piece behind the
        producer). With `src_split` the aggregation over the sources of piece k starts as soon as piece k has landed,
        i.e. while piece k + 1 is still on the links (one CSR per source piece, later pieces add to the rows of the
        earlier ones in a fixed order: reproducible) — measured on the emulated rank 0 of 8 this costs 0.7 ms of
        aggregation time per epoch at 2 pieces (half-length rows per launch, the output read again) for 1.9 ms less
        exposed exchange at 60 GB/s per link: the setting for slow links, off by default. Outbound: one send per target
        piece, in flight while the next one is aggregated.
        A generator: it YIELDS every exchange it is about to depend on (the caller waits — or first lets another forward
        enqueue its share, eval_pair) and returns `u`. `landed`: the caller has just come back from another yield, i.e.
        the inbound pieces have had a whole segment of someone else's work to travel: they are waited for directly
        instead of handing over once more after next to nothing."""
        R, C = self.shapes[i]
        d = self.specs[i].d_in
        dg, P = self.dg, self.P
        if inbound is None:
            cols, works = None, []
            for k in range(self.pieces_in):
                cols, work = self._inbound(i, blk, piece=(k, self.pieces_in), cols=cols)
                works.append(work)
            inbound = (cols, works)
        cols, works = inbound
        if not self.src_split and len(works) > 1:  # every inbound piece must have landed before the one aggregation
            works = [_AllOf(works)]
        src_pieces = len(works)
        half, handles = dg._grid_half(self.specs[i].kind, C, dg.pieces_for(d), direction, src_pieces)
        if src_pieces == 1:
            handles = [handles]
        dc = blk.size(2)
        u = self._blocked_buffer(i, width=C * dc)
        empty = u[0, 0:0]
        tag = f"dist_{direction}_colshard"
        ranges = [(half.piece_ptr[k], half.piece_ptr[k + 1]) for k in range(half.pieces)]
        sends = [cols.new_empty((hi - lo, dc)) for lo, hi in ranges]
        pending = []

        def send_piece(k):  # target piece k is complete: it leaves while the next one is aggregated
            a, b = half.my_piece[k]
            sv, rv, off = [], [], 0
            for q in range(P):
                cnt = half.piece_counts[k][q]
                sv.append(sends[k][off:off + cnt])
                off += cnt
                rv.append(u[q % C, a:b] if q in half.members else empty)
            pending.append(self.comm.all_to_all_views(sv, rv, tag=f"out {k + 1}/{half.pieces}" + self._pair_tag))

        if not all(self.be.rows_ok(h) for h in handles):
            # hub-row plan (row ids in it are absolute): whole-group launches once everything has landed. WHICH ranks
            # have such a plan depends on the graph: this branch must yield exactly where the other one does (the
            # interleave order of the generators, i.e. the order of the collectives, follows the yields)
            for work in works:
                if landed:
                    work.wait()
                else:
                    yield work
            full = self.be.run(handles[0], cols, tag)
            for h in handles[1:]:
                full.add_(self.be.run(h, cols, tag))
            sends = [full[a:b] for a, b in ranges]
            for k in range(half.pieces):
                send_piece(k)
        else:
            for ks in range(src_pieces):
                if landed:
                    works[ks].wait()
                else:
                    yield works[ks]
                for k, (lo, hi) in enumerate(ranges):
                    self.be.run_rows(handles[ks], cols, lo, hi, sends[k], tag, accumulate=ks > 0)
                    if ks == src_pieces - 1:
                        send_piece(k)
        for w in pending:  # `sends` stay referenced until their exchanges were waited on
            yield w
        return u

    # ---- training step -----------------------------------------------------------------------------------------
    @torch.no_grad()  # forward AND backward are written out here: nothing is recorded for autograd
    def train_step(self):
        """Forward + backward of one training step; leaves every parameter's .grad set (this rank's share) and
        returns this rank's share of the loss as a float64 device tensor [1]."""
        return _drive(self._train_g())

    def _bn_statistics_g(self, bn, h, cs, speculative):
        """B.train_statistics with the [2 d + 1] all-reduce issued asynchronously and YIELDED: another schedule's
        kernels can be enqueued before this one's stream starts to wait for it. A SPECULATIVE step (see
        eval_pair_and_next_step) moves copies of the running statistics; commit_speculation installs them."""
        packed = B.pack_statistics(h, cs)
        yield self.comm.all_reduce_sum_async(packed)
        if speculative:
            self._commit_if_due()
            rm, rv = bn.running_mean.clone(), bn.running_var.clone()
            mom = bn.momentum if bn.momentum is not None else 1.0 / (float(bn.num_batches_tracked) + 1.0)
            self._pending_bn.append((bn, rm, rv))
            running = (rm, rv, mom)
        else:
            running = bn.begin_training_step()
        return B.train_statistics(h, bn.weight, bn.bias, bn.eps, None, running, packed=packed)

    def accept_speculation(self):
        """The speculative step is THE next step after all (the caller takes its optimizer step): its running statistics
        become the model's — by three small copies that are enqueued where they cost nothing, i.e. behind the first
        large launch of the epoch's schedule and before anything reads the statistics (_commit_if_due)."""
        self._commit_list, self._pending_bn = self._pending_bn, []

    def note_optimizer_step(self):
        """The parameters changed (a fused Adam step moves no version counter): folded eval operands are stale."""
        self._opt_steps += 1

    def _commit_if_due(self):
        for bn, rm, rv in self._commit_list:
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            bn.num_batches_tracked += 1
        self._commit_list = []

    def discard_speculation(self):
        """The loop ended (or took another turn): nothing of the speculative step has touched the model."""
        self._pending_bn = []

    def _train_g(self, speculative=False):
        """train_step as a schedule generator (yields every exchange / all-reduce it is about to depend on)."""
        S, be = self.specs, self.be
        L = len(S)
        self._steps += 1  # the parameters are about to change: folded eval weights of the previous state are stale
        # (the module path keeps its own cache of eval operands, keyed by the model's count of training forwards)
        self.model._train_forwards = getattr(self.model, "_train_forwards", 0) + 1
        self._pending_bn = []
        wt = lambda w: w.detach().t().contiguous()
        s0 = S[0]
        blk, inbound, h, z, cs = self._first_layer(wt(s0.W), s0.bias(), None if s0.Wr is None else wt(s0.Wr), True)
        saved = [(z, None, None)]
        dl = None
        for i in range(1, L):
            sp, bn = S[i], self.bns[i - 1]
            stats = yield from self._bn_statistics_g(bn, h, cs, speculative)
            mean, rstd, scale, shift, n = stats
            # layer 1's inbound pieces were issued behind layer 0's launches and the statistics' all-reduce has just
            # yielded: they have landed; a deeper layer issues its exchange inside _propagate_g and yields it as usual
            u = yield from self._propagate_g(i, "fwd", blk, inbound, landed=inbound is not None)
            inbound = None
            pre = (scale, shift, self.rowsum())
            root = dict(x_root=h, wt_root=wt(sp.Wr)) if sp.Wr is not None else {}
            if i == L - 1:
                dl, z, ce = be.layer(u, wt(sp.W), bias=sp.bias(), pre=pre, want_z=True,
                                     ce=(self.y, self.masks[0], self.grad_scale), kind="return_linear_fwd", **root)
                saved.append((z, h, stats))
            else:
                blk = self._blocked_buffer(i + 1)
                h_next, z, cs = be.layer(u, wt(sp.W), bias=sp.bias(), pre=pre, want_z=True, want_colsums=True,
                                         out_blocked=blk, kind="return_linear_fwd", **root)
                saved.append((z, h, stats))
                h = h_next
        loss_part = (ce[0] / self.mask_counts[0]).reshape(1)
        # ---- backward
        dy = dl
        for i in range(L - 1, 0, -1):
            sp, bn = S[i], self.bns[i - 1]
            z, h_prev, (mean, rstd, scale, shift, n) = saved[i]
            # q = dy W goes out first; the weight-gradient GEMMs run while its slices travel
            q = self._blocked_buffer(i)
            be.layer(dy, sp.W.detach().contiguous(), want_out=False, out_blocked=q, kind="return_linear_bwd")
            cols, works = None, []
            for k in range(self.pieces_in):
                cols, work = self._inbound(i, q, piece=(k, self.pieces_in), cols=cols)
                works.append(work)
            _, gcol = be.gemm_tn(dy, z, colsum=True, out=self._g(sp.W), sums_out=self._g(sp.biases[0]))
            for b in sp.biases[1:]:
                self._g(b).copy_(gcol)
            if sp.Wr is not None:  # dWr = dy^T BN(h) = (dy^T h) diag(s) + colsum(dy) t^T
                torch.addcmul(gcol[:, None] * shift, be.gemm_tn(dy, h_prev), scale, out=self._g(sp.Wr))
            v = yield from self._propagate_g(i, "bwd", q, (cols, works))
            g_a = be.blocked_to_rows(v)
            if sp.Wr is not None:
                g_a.addmm_(dy, sp.Wr.detach())
            local = B.bwd_sums(g_a, h_prev, mean, rstd)
            glob = local.reshape(-1).clone()
            yield self.comm.all_reduce_sum_async(glob)
            dy, _, _ = B.train_backward(g_a, h_prev, bn.weight, mean, rstd, n, None,
                                        grads_out=(self._g(bn.weight), self._g(bn.bias)), sums=(local, glob))
        z0 = saved[0][0]
        _, gcol = be.gemm_tn(dy, z0, colsum=True, out=self._g(s0.W), sums_out=self._g(s0.biases[0]))
        for b in s0.biases[1:]:
            self._g(b).copy_(gcol)
        if s0.Wr is not None:
            be.gemm_tn(dy, self.x, out=self._g(s0.Wr))
        return loss_part

    # ---- eval forward ------------------------------------------------------------------------------------------
    @torch.no_grad()  # also called with autograd on (the pre-warm of the interleaved evals): nothing to record
    def _eval_weights(self):
        """Per layer (W'^T, b', Wr'^T) with the eval-mode BatchNorm behind the layer folded in (W' = diag(scale) W,
        b' = b scale + shift): made once per parameter state — the val and the test forward of an epoch share them."""
        # keyed by the training steps taken (a fused Adam step and the BatchNorm kernels write through raw pointers:
        # version counters do not see them) and, for changes from outside (load_state_dict), by the version counters
        from .. import ops
        self._commit_if_due()
        tensors = [p for p in self.model.parameters()] + [t for bn in self.bns for t in (bn.running_mean, bn.running_var)]
        key = (self._steps, self._opt_steps, ops.weights_epoch()) + tuple(t._version for t in tensors)
        if self._folded[0] != key:
            out = []
            L = len(self.specs)
            for i, sp in enumerate(self.specs):
                W, b, Wr = sp.W.detach(), sp.bias(), None if sp.Wr is None else sp.Wr.detach()
                if i < L - 1:
                    scale, shift = self.bns[i].eval_affine()
                    W, b = W * scale[:, None], b * scale + shift
                    Wr = None if Wr is None else Wr * scale[:, None]
                out.append([W, b, Wr])
            if self.last_transform_first:
                # eval mode, no root term: logits = P (a W_last^T) + b_last = P (a') + b_last with a' = a W_last^T, and
                # a = z W'^T + b' is itself linear in the previous aggregate, so the previous layer applies the product
                # W_last W' (and carries W_last b'); the last layer's return stage has no GEMM left
                W_last = out[L - 1][0]
                out[L - 2][1] = W_last @ out[L - 2][1]
                out[L - 2][0] = W_last @ out[L - 2][0]
                out[L - 1][0] = None
            out = [(None if W is None else W.t().contiguous(), b.contiguous(), None if Wr is None else Wr.t().contiguous())
                   for W, b, Wr in out]
            self._folded = (key, out)
        return self._folded[1]

    @torch.no_grad()
    def eval_stats(self, which):
        """[masked NLL sum, correct count] (float64 device tensor) of this rank's rows under masks[which], eval mode:
        every BatchNorm folded into the preceding layer's weights, no row-major activation written, no logits."""
        return _drive(self._eval_g(which))

    def _interleave(self, gens, tags, width=None):
        """Drive several schedule generators in turn on this one thread and stream: a generator runs until it yields the
        exchange (or all-reduce) it is about to depend on; the others then enqueue their kernels and exchanges up to THEIR
        next dependency before the first one's is waited for. The order depends only on the model and the scheme —
        identical on every rank, as the collectives of one communicator must be. `tags[i]` marks generator i's
        exchanges in the exchange log (bench.py). `width`: at most that many generators are in play at a time, the next
        one starts when one ends (a long schedule next to short ones: the short ones are then spread along it instead
        of all running beside its first part). Returns the generators' return values."""
        n = len(gens)
        width = n if width is None else width
        works, out, started = [None] * n, [None] * n, [False] * n
        active, waiting = list(range(min(width, n))), list(range(min(width, n), n))
        pos = 0
        while active:
            i = active[pos]
            self._pair_tag = tags[i]
            try:
                if works[i] is not None:
                    works[i].wait()
                works[i] = gens[i].send(None) if started[i] else next(gens[i])
                started[i] = True
                pos = (pos + 1) % len(active)
            except StopIteration as done:
                out[i] = done.value
                if waiting:
                    active[pos] = waiting.pop(0)  # the newcomer runs NEXT: it has a whole first segment to offer the
                else:                             # exchange the other generator has just issued
                    active.pop(pos)
                    pos = pos % len(active) if active else 0
            finally:
                self._pair_tag = ""
        return out

    @torch.no_grad()
    def eval_pair_and_next_step(self, first, second):
        """The val and the test forward of THIS epoch interleaved with the forward + backward of the NEXT epoch's
        training step (all three read the parameters the optimizer step of this epoch left; the next optimizer step is
        the caller's, once it knows the loop goes on — DistRunner.epoch). The training step's exchanges — above all the
        backward's inbound one, behind which a step on its own has nothing to run — travel while the eval forwards
        aggregate, and the other way round. Same kernels and numbers as eval_pair followed by train_step.
        The step is SPECULATIVE: it writes the gradient buffer and COPIES of the BatchNorm running statistics only, so
        the model is exactly the one the finished epoch left (state_dict, best-model snapshots, a loop that stops
        here: discard_speculation) until commit_speculation + the optimizer step make it the next step.
        Returns (stats of `first`, stats of `second`, loss share of the next step)."""
        # The training generator goes first: its layer-0 launches are in the queue before the small launches that fold
        # the eval operands (first thing the eval generators do) — an epoch that opens with twenty 5-microsecond launches
        # right after the host read-back runs them at the host's pace (measured: + 0.2 ms per epoch).
        # Two generators in play at a time: the first eval forward runs beside the training forward, the second one
        # beside the backward (all three at once would put both eval forwards beside the forward and leave the
        # backward's exchanges without company).
        t, a, b = self._interleave([self._train_g(speculative=True), self._eval_g(first), self._eval_g(second)],
                                   [" tri1", " tri2", " tri3"], width=2)
        return a, b, t

    @torch.no_grad()
    def eval_shared(self, first, second):
        """The val and the test statistics of an epoch from ONE eval forward (share_eval_forward: the reference's second
        eval forward, itexperiments.py:470, recomputes the very outputs of its first, :464): the last layer's return-stage
        launch takes both masks (rgbx_ce_epilogue_t.mask_groups = 2) — half the eval exchanges of eval_pair, the same
        four numbers. Returns ([nll sum, hits] under masks[first], the same under masks[second])."""
        st = _drive(self._eval_g((first, second)))
        return st[:2], st[2:]

    @torch.no_grad()
    def eval_shared_and_next_step(self, first, second):
        """eval_shared interleaved with the forward + backward of the NEXT epoch's training step (see
        eval_pair_and_next_step): two generators. Returns (stats first, stats second, loss share of the next step)."""
        t, st = self._interleave([self._train_g(speculative=True), self._eval_g((first, second))], [" tri1", " tri2"])
        return st[:2], st[2:], t

    @torch.no_grad()
    def eval_pair(self, first, second):
        """The val and the test forward of an epoch (itexperiments.py:464-473: two full eval forwards, both run),
        INTERLEAVED on one host thread and one stream: each forward is a generator that yields the exchange it is about
        to depend on; the other forward then enqueues its kernels and exchanges up to ITS next dependency before the
        first one's exchange is waited for. One forward's exchange is therefore in flight while the other's layer 0 /
        slice SpMM runs, in an order that depends only on the model and the scheme (identical on every rank), without a
        second thread (the module path's interleave costs 3.7 instead of 1.5 ms of host time per epoch, section 4.3).
        Same kernels and numbers as two forwards in a row."""
        return self._interleave([self._eval_g(first), self._eval_g(second)], [" paired1", " paired2"])

    def _eval_g(self, which, folded=None):
        """`which`: a mask index -> [nll sum, hits]; a PAIR of mask indices -> the four numbers [nll, hits, nll, hits] of
        both masks from this one forward (eval_shared)."""
        S, be = self.specs, self.be
        L = len(S)
        pair = isinstance(which, (tuple, list))
        prev_blk = inbound = None
        folded = self._eval_weights() if folded is None else folded
        for i in range(L):
            sp = S[i]
            wt, b, wtr = folded[i]
            if i == 0:
                prev_blk, inbound, _, _, _ = self._first_layer(wt, b, wtr, False)
                continue
            u = yield from self._propagate_g(i, "fwd", prev_blk, inbound)
            inbound = None
            root = dict(x_root=prev_blk, wt_root=wtr) if wtr is not None else {}
            if i == L - 1 and wt is None:  # the transform ran before the exchange: logits = aggregate + bias, and the
                if pair:  # statistics read the received slices in place (selected rows only), once per mask
                    return torch.cat([be.ce_stats_blocked(u, b, self.y, self.masks[w])[::2] for w in which])
                st = be.ce_stats_blocked(u, b, self.y, self.masks[which])
                return st[::2]
            if i == L - 1:
                mask = tuple(self.masks[w] for w in which) if pair else self.masks[which]
                _, _, st = be.layer(u, wt, bias=b, ce=(self.y, mask, None), kind="return_linear_fwd", **root)
                # (nll sum, hits) as a VIEW: indexing with a list would stage an index tensor through the host and
                # drain the queue (it did, twice per epoch, until round 3)
                return st[:, ::2].reshape(-1) if pair else st[::2]
            blk = self._blocked_buffer(i + 1, width=wt.size(1))
            be.layer(u, wt, bias=b, want_out=False, out_blocked=blk, kind="return_linear_fwd", **root)
            prev_blk = blk
