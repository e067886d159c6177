"""Autograd wrappers around the HIP aggregation kernels (C ABI: include/rgbx_hip.h).

Each Function is one ``MessagePassing.propagate`` of the reference: forward runs on the
target-grouped CSR, backward runs the SAME kernel on the source-grouped (transposed) CSR, so no
[E', d] tensor is ever materialised (cf. the gather -> multiply -> scatter temporaries of the
PyG path the reference uses, models/gcn.py:27).
"""
import ctypes

import torch

from . import _lib

# Optional per-launch timing hook used by bench.py: when set to a list, every aggregation launch
# appends (kind, start_event, end_event) recorded on the launch stream.
_EVENT_SINK = None


def set_event_sink(sink):
    global _EVENT_SINK
    _EVENT_SINK = sink


class Kind(str):
    """Event kind that also says WHICH FORM of the launch it was (`variant`: the epilogues / extra outputs that select
    the kernel's template instantiation and add to its bytes); equal to, and hashed as, the plain kind string."""
    variant = None


class _Timed:
    def __init__(self, kind, variant=None):
        if variant is not None and _EVENT_SINK is not None:
            kind = Kind(kind)
            kind.variant = variant
        self.kind = kind

    def __enter__(self):
        if _EVENT_SINK is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if _EVENT_SINK is not None:
            self.e.record()
            _EVENT_SINK.append((self.kind, self.s, self.e))
        return False


def spmm_raw(csr, w, rs, x, y=None, a=1.0, b=0.0, out=None, kind="spmm", bias=None):
    """out[i] = a*rs[i]*sum_p w[p]*x[col[p]] + b*y[i] + bias over `csr` (no autograd)."""
    _lib.require_device(x, y, bias)
    x = x if x.stride(-1) == 1 else x.contiguous()
    px, ldx = _lib.mat(x, "x")
    N, d = csr.N, x.size(1)
    if out is None:
        out = torch.empty((N, d), dtype=torch.float32, device=x.device)
    po, ldo = _lib.mat(out, "out")
    py, ldy = (0, 0) if y is None else _lib.mat(y, "y")
    split, _scratch = csr.split_arg(d, x.device)
    with _Timed(kind, f"rows+d{d}" if _EVENT_SINK is not None else None):
        _lib.check(
            _lib.load().rgbx_spmm_csr_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(w), _lib.ptr(rs),
                                          px, ldx, py, ldy, _lib.ptr(bias), po, ldo, N, d, float(a), float(b),
                                          None if split is None else ctypes.byref(split), _lib.stream_ptr()),
            "rgbx_spmm_csr_f32")
    return out


class _PropagateGCN(torch.autograd.Function):
    """A_hat x with A_hat = D^-1/2 (A + I) D^-1/2 (dagnn.py:12-31, message dagnn.py:57-59)."""

    @staticmethod
    def forward(ctx, x, graph, bias):
        ctx.graph, ctx.has_bias = graph, bias is not None
        b = None if bias is None else bias.detach().contiguous()
        return spmm_raw(graph.fwd, graph.w, None, x, kind="gcn_fwd", bias=b)

    @staticmethod
    def backward(ctx, gy):
        g = ctx.graph
        gy = gy.contiguous()
        gb = gy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        gx = spmm_raw(g.bwd, g.w_t, None, gy, kind="gcn_bwd") if ctx.needs_input_grad[0] else None
        return gx, None, gb


class _PropagateMean(torch.autograd.Function):
    """mean_{j in N(i)} x_j with sum/max(count,1) (aggr='mean', graphsage.py:39,58)."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return spmm_raw(graph.fwd, None, graph.inv_deg, x, kind="mean_fwd")

    @staticmethod
    def backward(ctx, gy):
        g = ctx.graph
        return spmm_raw(g.bwd, g.w_mean_t, None, gy.contiguous(), kind="mean_bwd"), None


class _PropagateSum(torch.autograd.Function):
    """Unweighted sum over incoming edges (aggr='add', no norm)."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return spmm_raw(graph.fwd, None, None, x, kind="sum_fwd")

    @staticmethod
    def backward(ctx, gy):
        return spmm_raw(ctx.graph.bwd, None, None, gy.contiguous(), kind="sum_bwd"), None


def _is_dist(graph):
    return getattr(graph, "is_distributed", False)


def target_rows(x, graph):
    """The rows of `x` that are aggregation TARGETS of `graph`: all of them, except on a partitioned run's
    rectangular second stage (dist.ReplicaGraph: sources = all N replicated rows, targets = this rank's)."""
    fn = getattr(graph, "target_rows", None)
    return x if fn is None else fn(x)


def align_rows(x):
    """`x` ([N, F] float32 on the device) as a view whose ROWS start on 16-byte boundaries: F % 4 != 0 (Cora's 1433) ->
    a fresh [N, F4] buffer (F4 = F rounded up to 4, pad columns zero) viewed as [:, :F]. Same values, same shape; dense
    products over it read whole float4s (rgbx_gemm_tn_f32's aligned-row path, hipBLASLt with lda = F4) and _pad4 finds the
    padded matrix already there. Costs one copy, once per feature matrix (the reference moves / normalises the features
    once as well, itexperiments.py:258-299); returns `x` itself where nothing is to do."""
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.size(1) % 4 == 0:
        return x
    n, f = x.shape
    base = torch.zeros((n, (f + 3) // 4 * 4), dtype=torch.float32, device=x.device)
    view = base[:, :f]
    view.copy_(x)
    view._rgbx_base = base  # (this Python object only: views of it are ordinary strided tensors)
    return view


def _aligned_rows(t):
    """`t` itself when its rows sit on 16-byte boundaries (or nothing would be gained), else align_rows' padded copy."""
    if t.dim() != 2 or t.dtype != torch.float32 or t.size(1) % 4 == 0 or (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
        return t
    return align_rows(t)


def _rows_padded_readable(t):
    """A [K, n] matrix whose 16-byte-aligned rows may be read up to the next multiple of 4 columns (rgbx_gemm_tn_f32's
    contract): contiguous rows, stride % 4 == 0, and the storage holds the last row's padding too."""
    if t.dim() != 2 or t.size(1) % 4 == 0 or t.stride(0) % 4 or t.data_ptr() % 16:
        return True  # nothing is read beyond the columns (or the rows are unaligned: the 4-byte path)
    last = t.storage_offset() + (t.size(0) - 1) * t.stride(0) + (t.size(1) + 3) // 4 * 4
    return last * 4 <= t.untyped_storage().nbytes()


def _pad4(x):
    """Feature widths that are no multiple of 4 (C = 7 classes on Cora, 41 on Reddit, 47 on ogbn-products) run on
    zero-padded rows: 16-byte aligned rows take the float4 gather path and straddle fewer 128-byte lines. Measured at
    |V| = 2M, |E| = 60M: d = 7 1.22 ms vs 8 1.08 ms, 41 2.46 vs 44/48 2.21, 47 2.64 vs 48 2.21 (profiles/
    r02_narrow_width_ms.txt); the pad copy is N x d floats against E' x d gathered. Returns (x or its padded copy,
    original width); zero columns aggregate to zero columns, the caller slices them off (autograd pads the gradient)."""
    d = x.size(1)
    if d % 4 == 0 or not x.is_cuda:
        return x, d
    base = getattr(x, "_rgbx_base", None)
    if base is not None and not x.requires_grad:  # align_rows made the padded matrix already (pad columns zero)
        return base, d
    return torch.nn.functional.pad(x, (0, 4 - d % 4)), d


def _pad4_vec(v, d_padded):
    return v if v is None or v.numel() == d_padded else torch.nn.functional.pad(v, (0, d_padded - v.numel()))


def propagate_gcn(x, graph, bias=None):
    """A_hat x (+ bias, fused into the kernel's store)."""
    if _is_dist(graph):
        out = graph.propagate(x, "gcn")
        return out if bias is None else out + bias
    xp, d = _pad4(x)
    out = _PropagateGCN.apply(xp, graph, _pad4_vec(bias, xp.size(1)))
    return out if xp is x else out[:, :d]


def propagate_mean(x, graph):
    if _is_dist(graph):
        return graph.propagate(x, "mean")
    xp, d = _pad4(x)
    out = _PropagateMean.apply(xp, graph)
    return out if xp is x else out[:, :d]


def propagate_sum(x, graph):
    if _is_dist(graph):
        return graph.propagate(x, "sum")
    xp, d = _pad4(x)
    out = _PropagateSum.apply(xp, graph)
    return out if xp is x else out[:, :d]


# ---- propagate of an already transformed matrix, with the passes that follow it taken in the same kernel --------------

def pad_rows4(weight, *vectors):
    """A Linear's weight [out, in] (and [out] vectors next to it) with zero rows up to the next multiple of 4: the product
    then comes out in 16-byte rows (C = 7 classes -> 8 columns) and no pad copy of the [N, C] matrix is needed in front of
    the aggregation (cf. _pad4). Differentiable (the pad's backward is a slice). Returns (weight', *vectors', padded out)."""
    n = weight.size(0)
    p = (-n) % 4
    if p == 0:
        return (weight,) + vectors + (n,)
    pad = torch.nn.functional.pad
    return (pad(weight, (0, 0, 0, p)),) + tuple(None if v is None else pad(v, (0, p)) for v in vectors) + (n + p,)


def rows_epilogue_ok(graph, width, x=None, y=None):
    """The row gather can take BatchNorm's column sums / the masked cross-entropy of its output into the kernel
    (rgbx_spmm_csr_epilogue_f32): single-GPU graph, device tensors, `width` (a multiple of 4) <= 256."""
    return (not _is_dist(graph) and width % 4 == 0 and 0 < width <= 256 and (x is None or x.is_cuda)
            and (y is None or (y.is_cuda and y.dtype == torch.int64)))


def spmm_epilogue_raw(csr, w, rs, x, y=None, a=1.0, b=0.0, bias=None, out=None, want_colsums=False, ce=None, n_classes=0,
                      kind="spmm"):
    """spmm_raw with an epilogue over the finished rows (no autograd; rgbx_spmm_csr_epilogue_f32). Exactly one of
    `want_colsums` -> (out, colsums [2, d] float64 of out and out^2), and `ce` = (labels, mask or (mask_a, mask_b),
    grad_scale or None) -> (loss gradient w.r.t. the logits, or None when grad_scale is None: nothing is written; stats [3]
    or [2, 3] float64 = nll sum, selected rows, correct). `n_classes`: columns beyond are padding (see pad_rows4)."""
    _lib.require_device(x, y, bias)
    lib = _lib.load()
    px, ldx = _lib.mat(x, "x")
    N, d = csr.N, x.size(1)
    py, ldy = (0, 0) if y is None else _lib.mat(y, "y")
    keep = []
    E = _lib.SpmmEpilogue()
    stats = colsums = None
    want_out = True
    if ce is not None:
        labels, mask, grad_scale = ce
        groups = 1
        if isinstance(mask, (tuple, list)):
            if grad_scale is not None:
                raise RuntimeError("spmm_epilogue_raw: two masks are for statistics only (no loss gradient)")
            mask, groups = group_masks(*mask), 2
        _lib.require_device(labels, mask, grad_scale)
        if labels.dtype != torch.int64:
            raise RuntimeError(f"labels must be int64, got {labels.dtype}")
        if mask is not None and mask.dtype not in (torch.bool, torch.uint8):
            raise RuntimeError(f"mask must be bool, got {mask.dtype}")
        labels = labels.contiguous()
        mask = None if mask is None else mask.contiguous()
        stats = torch.empty(3 * groups, dtype=torch.float64, device=x.device)
        scratch = torch.empty(3 * groups * ((N + 31) // 32 + 64), dtype=torch.float64, device=x.device)
        ce_arg = _lib.CeEpilogue(_lib.ptr(labels), _lib.ptr(mask), _lib.ptr(grad_scale), _lib.ptr(stats), _lib.ptr(scratch),
                                 groups)
        keep += [labels, mask, scratch, ce_arg, grad_scale]
        E.ce, E.n_classes = ctypes.addressof(ce_arg), int(n_classes)
        want_out = grad_scale is not None
    elif want_colsums:
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.rgbx_spmm_linear_stats_workspace_bytes(N, d, ctypes.byref(nbytes)), "rgbx_spmm_linear_stats_workspace_bytes")
        ws = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=x.device)
        colsums = torch.empty((2, d), dtype=torch.float64, device=x.device)
        keep.append(ws)
        E.out_colsums, E.stats_ws, E.stats_ws_bytes = colsums.data_ptr(), ws.data_ptr(), nbytes.value
    else:
        raise RuntimeError("spmm_epilogue_raw: no epilogue asked for (use spmm_raw)")
    if out is None and want_out:
        out = torch.empty((N, d), dtype=torch.float32, device=x.device)
    po, ldo = (0, d) if out is None else _lib.mat(out, "out")
    split, _scratch = csr.split_arg(d, x.device, hub_rows=1)
    variant = None
    if _EVENT_SINK is not None:
        variant = "rows+" + ("stats" if ce is None else ("ce_grad" if want_out else "ce_stats")) + f"+d{d}"
    with _Timed(kind, variant):
        _lib.check(
            lib.rgbx_spmm_csr_epilogue_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(w), _lib.ptr(rs), px, ldx,
                                           py, ldy, _lib.ptr(bias), po, ldo, N, d, float(a), float(b),
                                           None if split is None else ctypes.byref(split), ctypes.byref(E),
                                           _lib.stream_ptr()), "rgbx_spmm_csr_epilogue_f32")
    del keep
    if ce is not None:
        return out, (stats.view(2, 3) if stats.numel() == 6 else stats)
    return out, colsums


def _kind_weights(graph, kind, transposed=False):
    """(per-slot weights, per-row scale) of the aggregation `kind` on the forward / the transposed CSR."""
    if kind == "gcn":
        return (graph.w_t, None) if transposed else (graph.w, None)
    if kind == "mean":
        return (graph.w_mean_t, None) if transposed else (None, graph.inv_deg)
    if kind == "sum":
        return None, None
    raise ValueError(kind)


class _PropagateRows(torch.autograd.Function):
    """out = P h[:, :n] (+ h[:, n:2n]) (+ bias), P = A_hat ('gcn'), the mean operator ('mean') or the plain edge sum
    ('sum'): the aggregation of a layer that has ALREADY been transformed (in > out: the reference's default shapes,
    initial_params.py:25-29), with the layer's root / self term as the right half of the same matrix — one GEMM
    x [W_l; W_r]^T reads the (wide) input once for both halves (models/graphsage.py:49-50 runs two Linears, SAGEConv [PyG]
    likewise) and the add is the kernel's `y` operand. `mode`:
      'plain'    returns out                      (rgbx_spmm_csr_f32)
      'colsums'  returns (out, colsums [2, n])    (a training-mode BatchNorm follows: models/gcn.py:28)
      'ce'       returns (loss, stats)            (the model's last layer: gcn.py:31 + itexperiments.py:429,624-626);
                 ce_args = (labels, mask, n_classes, want_grad); the logits are never written
    Backward: g_h[:, :n] = P^T g_out on the transposed CSR, g_h[:, n:] = g_out, g_bias = column sums of g_out; in 'ce'
    mode g_out is the loss gradient the forward stored, times the incoming scalar."""

    @staticmethod
    def forward(ctx, h, graph, kind, n, bias, mode, ce_args):
        split = h.size(1) == 2 * n
        if not split and h.size(1) != n:
            raise RuntimeError(f"propagate_rows: h is {tuple(h.shape)} for n = {n}")
        h = h if h.stride(-1) == 1 else h.contiguous()
        w, rs = _kind_weights(graph, kind)
        x = h[:, :n] if split else h
        y = h[:graph.fwd.N, n:] if split else None
        b = None if bias is None else bias.detach().contiguous()
        prefix = getattr(graph, "event_prefix", "")
        ctx.graph, ctx.kind, ctx.n, ctx.split, ctx.mode = graph, kind, n, split, mode
        ctx.has_bias, ctx.h_shape = bias is not None, tuple(h.shape)
        if mode == "plain":
            return spmm_raw(graph.fwd, w, rs, x, y=y, a=1.0, b=1.0, kind=f"{prefix}{kind}_fwd", bias=b)
        if mode == "colsums":
            out, cs = spmm_epilogue_raw(graph.fwd, w, rs, x, y=y, a=1.0, b=1.0, bias=b, want_colsums=True,
                                        kind=f"{prefix}{kind}_fwd")
            ctx.mark_non_differentiable(cs)
            return out, cs
        labels, mask, n_classes, want_grad = ce_args
        grad_scale = mask_scale(labels, mask, n_classes) if want_grad else None
        dlogits, stats = spmm_epilogue_raw(graph.fwd, w, rs, x, y=y, a=1.0, b=1.0, bias=b,
                                           ce=(labels, mask, grad_scale), n_classes=n_classes, kind=f"{prefix}{kind}_fwd")
        if want_grad:
            ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(stats)
        if stats.dim() == 2:  # two masks, statistics only
            return stats.new_zeros(()).float(), stats
        return (stats[0] / stats[1]).float(), stats

    @staticmethod
    def backward(ctx, g, _g_extra=None):
        graph, kind, n = ctx.graph, ctx.kind, ctx.n
        if ctx.mode == "ce":
            if not ctx.saved_tensors:
                raise RuntimeError("propagate_rows: backward asked of a loss forward that was run without want_grad")
            gy = ctx.saved_tensors[0] * g.reshape(()).float()
        else:
            gy = g.contiguous()
        gh = gb = None
        if ctx.needs_input_grad[0]:
            wt, _ = _kind_weights(graph, kind, transposed=True)
            prefix = getattr(graph, "event_prefix", "")
            gh = torch.empty(ctx.h_shape, dtype=torch.float32, device=gy.device)
            if ctx.split:
                gh[:graph.fwd.N, n:].copy_(gy)
                if gh.size(0) > graph.fwd.N:
                    gh[graph.fwd.N:, n:].zero_()
            spmm_raw(graph.bwd, wt, None, gy, out=gh[:, :n] if ctx.split else gh, kind=f"{prefix}{kind}_bwd")
        if ctx.has_bias and ctx.needs_input_grad[4]:
            gb = gy.sum(0)
        return gh, None, None, None, gb, None, None


def propagate_rows(h, graph, kind, n=None, bias=None, want_colsums=False):
    """P h[:, :n] (+ h[:, n:2n]) (+ bias) — see _PropagateRows; `n` defaults to h's width (no addend half). With
    `want_colsums` the output carries its column sums for the BatchNorm that follows (ops.COLSUMS) where the kernel can
    take them (rows_epilogue_ok), and is a plain output otherwise."""
    n = h.size(1) if n is None else n
    if want_colsums and rows_epilogue_ok(graph, n, h):
        return _tag_colsums(_PropagateRows.apply(h, graph, kind, n, bias, "colsums", None), True)
    return _PropagateRows.apply(h, graph, kind, n, bias, "plain", None)


def propagate_rows_ce(h, graph, kind, n, n_classes, y, mask, bias=None):
    """(loss, stats) of the masked cross-entropy of the logits P h[:, :n] (+ h[:, n:2n]) (+ bias), columns [n_classes, n)
    being padding — taken inside the gather kernel (the caller checked rows_epilogue_ok). `mask` = (mask_a, mask_b):
    (None, [2, 3] statistics) of one eval forward under both masks."""
    pair = isinstance(mask, (tuple, list))
    want_grad = (not pair) and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (h, bias))
    loss, stats = _PropagateRows.apply(h, graph, kind, n, bias, "ce", (y, mask, int(n_classes), want_grad))
    return (None if pair else loss), stats


def fused_linear_ok(graph, in_channels, out_channels, root=False, x=None):
    """The fused aggregate-then-transform kernel applies: supported widths, aggregating at the input width is
    not the more expensive order, and the graph is a single-GPU one — or a partitioned one asked to propagate its
    resident input features `x` (dist.DistGraph.pin_resident), which needs no exchange. `root`: with the
    SAGE-style root term x_i Wr^T accumulated in the same kernel."""
    if in_channels > out_channels:
        return False
    if _is_dist(graph):
        return x is not None and graph.fused_resident_ok(x, in_channels, out_channels, root)
    return bool(_lib.load().rgbx_spmm_linear_supported(in_channels, out_channels, int(root)))


_WEIGHTS_EPOCH = [0]


def note_weights_changed(*_args, **_kwargs):
    """Parameters may have been written without their Python-side version counters moving: a fused / foreach /
    capturable optimizer step (torch.optim.Adam(fused=True) writes through raw pointers), the replay of a captured
    hipGraph. Everything this package caches per parameter state (weight_t, ConvStack._eval_operands,
    dist.stack's folded operands) carries this counter in its key, so nothing made before the write is served after
    it. Registered below as a GLOBAL optimizer step post-hook: every torch.optim step of the process bumps it,
    whoever built the optimizer — a module of this package inside a foreign training loop needs no cooperation."""
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch():
    return _WEIGHTS_EPOCH[0]


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_post_hook  # noqa: E402

_register_step_post_hook(note_weights_changed)


class _MadeOn:
    """A device tensor kept for reuse, with the stream it was made on. Whoever takes it on ANOTHER stream first makes that
    stream wait for the making — the interleaved eval forwards of a multi-rank run (dist/runner.py, one HIP stream per
    forward, two host threads) share W^T copies, mask bitmaps and loss divisors: the forward that finds one cached must not
    launch on it before the other stream's transpose kernel has run. (Found by the first RCCL run with the ranks sharing a
    GPU, round 4: the replicate scheme's test loss, second forward, came out of the PREVIOUS step's W^T in the recycled block.)"""
    __slots__ = ("value", "stream", "event")

    def __init__(self, value):
        self.value, self.stream, self.event = value, None, None
        if value.is_cuda and not torch.cuda.is_current_stream_capturing():
            self.stream = torch.cuda.current_stream(value.device)
            self.event = torch.cuda.Event()
            self.event.record(self.stream)

    def get(self):
        if self.event is not None:
            cur = torch.cuda.current_stream(self.value.device)
            if cur != self.stream and not torch.cuda.is_current_stream_capturing():
                cur.wait_event(self.event)
                self.value.record_stream(cur)  # when the copy is replaced, its block is not handed out under this stream
        return self.value


def weight_t(weight):
    """W^T as the contiguous [K, Nout] operand rgbx_spmm_linear_f32 reads (its B fragments run along Nout). For an
    nn.Parameter the transposed copy is kept on the parameter and reused until the parameter may have changed: the
    in-place version counter, the storage address AND weights_epoch() (every optimizer step of the process, every
    hipGraph replay of epoch_graph) are its tag, instead of one `.t().contiguous()` per launch; temporaries (weights
    with a folded BatchNorm) and launches being captured into a hipGraph always transpose. Not seen: writes through
    `param.data` outside an optimizer (a fresh version counter by PyTorch's rules) — call note_weights_changed()."""
    w = weight.detach()
    if not isinstance(weight, torch.nn.Parameter) or (w.is_cuda and torch.cuda.is_current_stream_capturing()):
        return w.t().contiguous()
    tag = (weight._version, _WEIGHTS_EPOCH[0], w.data_ptr(), tuple(w.shape))
    cached = getattr(weight, "_rgbx_wt", None)
    if cached is None or cached[0] != tag:
        cached = (tag, _MadeOn(w.t().contiguous()))
        weight._rgbx_wt = cached
    return cached[1].get()


def _blocked(t, what):
    """(pointer, block columns, block stride) of a blocked matrix: a [B, n, cols] tensor whose blocks are contiguous
    [n, cols] matrices (views over the row range of a bigger one keep the block stride of their base)."""
    if (t.dtype != torch.float32 or t.dim() != 3 or (t.size(2) > 1 and t.stride(2) != 1)
            or (t.size(1) > 1 and t.stride(1) != t.size(2))):  # strides of size-1 dimensions mean nothing
        raise RuntimeError(f"{what}: expected a float32 [blocks, rows, cols] tensor with contiguous blocks, got "
                           f"{t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), t.size(2), t.stride(0)


def fused_layer(x, wt, csr=None, w=None, rs=None, bias=None, x_root=None, wt_root=None, pre=None, want_out=True,
                out_blocked=None, want_z=False, want_colsums=False, ce=None, kind="linear", out=None, z=None, w_pos=None):
    """One conv layer's arithmetic on rgbx_fused_layer_f32 (no autograd):
        z   = rs * sum_p w_p x[col_p]   over `csr`            (csr given: aggregate)
            = x                                                 (csr None: DENSE mode, the rows are loaded)
        z   = z * scale + shift * rowsum                        (`pre` = (scale [K], shift [K], rowsum [N]))
        out = z wt + bias (+ x_root' wt_root),  x_root' = x_root * scale + shift
    `x` is [n_src, K] row-major or, in DENSE mode, possibly BLOCKED ([B, N, K / B] — column slices as the exchange of
    a partitioned run delivers them); `x_root` likewise. `out_blocked` ([B', N, Nout / B'], optional): the output is
    (also) written there, in the layout the exchange sends from. `want_out=False`: no row-major output.
    `want_z`: the (mapped) aggregate is stored. `want_colsums`: [2, Nout] float64 column sums of out and out^2.
    `ce` = (y, mask, grad_scale): loss epilogue (see spmm_linear_raw). `out` / `z`: caller's row-major [N, Nout] /
    [N, K] tensors to write into (row ranges of bigger ones when a layer is launched piece by piece).
    `w_pos` (second weight vector over the same slots; implies want_z): z_pos = sum_p w_pos_p x[col_p] is stored as well
    and the return value is (out, (z, z_pos), colsums or stats).
    Returns (out, z, colsums or stats)."""
    _lib.require_device(x, wt, bias, x_root, wt_root, out_blocked)
    lib = _lib.load()
    L = _lib.FusedLayer()
    dense = csr is None
    if x.dim() == 3:
        if not dense:
            raise RuntimeError("fused_layer: a blocked x needs DENSE mode (no csr)")
        L.x, L.x_blk_cols, L.x_blk_stride = _blocked(x, "x")
        n_rows, K = x.size(1), x.size(0) * x.size(2)
    else:
        x = x if x.stride(-1) == 1 else x.contiguous()
        L.x, L.ldx = x.data_ptr(), x.stride(0) if x.size(0) > 1 else x.size(1)
        n_rows, K = x.size(0), x.size(1)
    N = n_rows if dense else csr.N
    wt = wt.contiguous()
    n_out = wt.size(1)
    if wt.size(0) != K:
        raise RuntimeError(f"fused_layer: wt is {tuple(wt.shape)}, the input width is {K}")
    keep = [x, wt]
    L.wt = wt.data_ptr()
    if not dense:
        L.rowptr, L.col, L.w, L.rs = _lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(w), _lib.ptr(rs)
        split, _scratch = csr.split_arg(K, x.device, hub_rows=2 if w_pos is not None else 1)
        keep.append(_scratch)
        if split is not None:
            keep.append(split)
            L.split = ctypes.addressof(split)
    if wt_root is not None:
        wtr = wt_root.contiguous()
        keep.append(wtr)
        L.wt_root = wtr.data_ptr()
        if x_root.dim() == 3:
            L.x_root, L.xr_blk_cols, L.xr_blk_stride = _blocked(x_root, "x_root")
        else:
            xr = x_root if x_root.stride(-1) == 1 else x_root.contiguous()
            keep.append(xr)
            L.x_root, L.ldr = xr.data_ptr(), xr.stride(0) if xr.size(0) > 1 else xr.size(1)
    if bias is not None:
        b = bias.contiguous()
        keep.append(b)
        L.bias = b.data_ptr()
    if pre is not None:
        ps, pt, pr = (t.contiguous() for t in pre)
        keep += [ps, pt, pr]
        L.pre_scale, L.pre_shift, L.pre_rowsum = ps.data_ptr(), pt.data_ptr(), pr.data_ptr()
    ce_stats = None
    groups = 1
    if ce is not None:
        y, mask, grad_scale = ce
        if isinstance(mask, (tuple, list)):  # two masks, one forward: statistics set 0 / 1 = bit 0 / 1 of the grouped mask
            if grad_scale is not None:
                raise RuntimeError("fused_layer: two masks are for statistics only (no loss gradient)")
            mask, groups = group_masks(*mask), 2
        _lib.require_device(y, mask, grad_scale)
        if y.dtype != torch.int64:
            raise RuntimeError(f"labels must be int64, got {y.dtype}")
        if mask is not None and mask.dtype not in (torch.bool, torch.uint8):
            raise RuntimeError(f"mask must be bool, got {mask.dtype}")
        y = y.contiguous()
        mask = None if mask is None else mask.contiguous()
        ce_stats = torch.empty(3 * groups, dtype=torch.float64, device=x.device)
        ce_scratch = torch.empty(3 * groups * ((N + 31) // 32 + 64), dtype=torch.float64, device=x.device)
        ce_arg = _lib.CeEpilogue(_lib.ptr(y), _lib.ptr(mask), _lib.ptr(grad_scale), _lib.ptr(ce_stats),
                                 _lib.ptr(ce_scratch), groups)
        keep += [y, mask, ce_scratch, ce_arg]
        L.ce = ctypes.addressof(ce_arg)
        want_out = grad_scale is not None  # statistics only: the logits are never written
    if out is not None:
        if tuple(out.shape) != (N, n_out) or out.stride(1) != 1 or out.dtype != torch.float32:
            raise RuntimeError(f"fused_layer: out is {tuple(out.shape)}, expected a float32 [{N}, {n_out}] with contiguous rows")
    elif want_out:
        out = torch.empty((N, n_out), dtype=torch.float32, device=x.device)
    if out is not None:
        L.out, L.ldo = out.data_ptr(), out.stride(0) if N > 1 else n_out
    else:
        L.ldo = n_out
    if out_blocked is not None:
        L.out_blk, L.ob_cols, L.ob_stride = _blocked(out_blocked, "out_blocked")
        if out_blocked.size(1) != N or out_blocked.size(0) * out_blocked.size(2) != n_out:
            raise RuntimeError(f"fused_layer: out_blocked is {tuple(out_blocked.shape)} for a [{N}, {n_out}] output")
    if z is not None:
        if tuple(z.shape) != (N, K) or z.stride(1) != 1 or z.dtype != torch.float32:
            raise RuntimeError(f"fused_layer: z is {tuple(z.shape)}, expected a float32 [{N}, {K}] with contiguous rows")
    elif want_z or w_pos is not None:
        z = torch.empty((N, K), dtype=torch.float32, device=x.device)
    L.ldz = K
    if z is not None:
        L.z_out, L.ldz = z.data_ptr(), z.stride(0) if N > 1 else K
    z_pos = None
    if w_pos is not None:
        _lib.require_device(w_pos)
        z_pos = torch.empty_strided((N, K), (L.ldz, 1), dtype=torch.float32, device=x.device)
        keep.append(w_pos)
        L.w_pos, L.z_pos_out = w_pos.data_ptr(), z_pos.data_ptr()
    colsums = None
    if want_colsums:
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.rgbx_spmm_linear_stats_workspace_bytes(N, n_out, ctypes.byref(nbytes)),
                   "rgbx_spmm_linear_stats_workspace_bytes")
        ws = torch.empty(max(nbytes.value, 8), dtype=torch.uint8, device=x.device)
        colsums = torch.empty((2, n_out), dtype=torch.float64, device=x.device)
        keep.append(ws)
        L.out_colsums, L.stats_ws, L.stats_ws_bytes = colsums.data_ptr(), ws.data_ptr(), nbytes.value
    L.N, L.K, L.Nout = N, K, n_out
    variant = None
    if _EVENT_SINK is not None:  # which form this launch is (bench.py's per-variant table)
        variant = "+".join([name for name, on in (
            ("dense", dense), ("pre", pre is not None), ("root", wt_root is not None), ("z", z is not None),
            ("zpos", w_pos is not None), ("stats", want_colsums), ("ce_stats", ce is not None and ce[2] is None),
            ("ce_grad", ce is not None and ce[2] is not None), ("noout", out is None and out_blocked is None and ce is None),
            ("blk", out_blocked is not None)) if on]) or "plain"
    with _Timed(kind, variant):
        _lib.check(lib.rgbx_fused_layer_f32(ctypes.byref(L), _lib.stream_ptr()), "rgbx_fused_layer_f32")
    del keep
    if groups == 2:
        ce_stats = ce_stats.view(2, 3)  # row k = [nll sum, selected rows, correct] under mask k
    return out, (z if w_pos is None else (z, z_pos)), (ce_stats if ce is not None else colsums)


def blocked_to_rows(src, out=None, bias=None):
    """[B, n, cols] blocked -> [n, B * cols] row-major (+ bias per column) (rgbx_blocked_to_rows_f32)."""
    _lib.require_device(src, bias)
    ptr, bc, bs = _blocked(src, "src")
    n, d = src.size(1), src.size(0) * src.size(2)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=src.device)
    b = None if bias is None else bias.detach().contiguous()
    _lib.check(_lib.load().rgbx_blocked_to_rows_f32(ptr, bc, bs, out.data_ptr(), out.stride(0), n, d, _lib.ptr(b),
                                                    _lib.stream_ptr()), "rgbx_blocked_to_rows_f32")
    return out


def spmm_linear_raw(csr, w, rs, x, wt, bias=None, want_z=False, x_root=None, wt_root=None, kind="linear", pre=None,
                    want_colsums=False, ce=None):
    """out = (rs * sum_p w_p x[col_p]) wt + bias (+ x_root wt_root) on the fused aggregate+transform kernel; `wt` /
    `wt_root` are [K, Nout] row-major. Returns (out, z) with z the stored aggregate [N, K] if `want_z`. `pre` = (scale
    [K], shift [K], rowsum [N]): the gathered matrix (and the root rows) stand for x * scale + shift. `want_colsums`:
    returns (out, z, colsums) with colsums [2, Nout] float64 = column sums of out and out^2 from the MFMA tiles.
    `ce` = (y, mask, grad_scale): the layer is the model's last and its logits go straight into the masked
    cross-entropy (rgbx_ce_epilogue_t): returns (out, z, stats) with stats [3] float64 = (nll sum, selected rows,
    correct); out = the loss gradient grad_scale * (softmax - onehot) when grad_scale is a device scalar, None (nothing
    written) when it is None."""
    out, z, extra = fused_layer(x, wt, csr=csr, w=w, rs=rs, bias=bias, x_root=x_root if wt_root is not None else None,
                                wt_root=wt_root, pre=pre, want_z=want_z, want_colsums=want_colsums, ce=ce, kind=kind)
    if ce is not None or want_colsums:
        return out, z, extra
    return out, z


class _AggregateLinear(torch.autograd.Function):
    """y = z W^T + b (+ x_root Wr^T) for an aggregate z = P x that was formed earlier and is kept (the opt-in
    cache of the static input features' aggregate, models/_stack.ConvStack.cache_input_aggregate): the DENSE form of
    rgbx_fused_layer_f32. z and x_root take no gradient (they are functions of the input features only);
    dW = dy^T z, db = column sums of dy, dWr = dy^T x_root."""

    @staticmethod
    def forward(ctx, z, weight, bias, root_weight, x_root, want_colsums):
        out, _, cs = fused_layer(z, weight_t(weight), bias=None if bias is None else bias.detach(),
                                 x_root=x_root if root_weight is not None else None,
                                 wt_root=None if root_weight is None else weight_t(root_weight),
                                 want_colsums=want_colsums, kind="cached_aggregate_linear_fwd")
        ctx.save_for_backward(z, x_root if root_weight is not None else None)
        ctx.has_bias = bias is not None
        if want_colsums:
            ctx.mark_non_differentiable(cs)
            return out, cs
        return out

    @staticmethod
    def backward(ctx, gy, _g_cs=None):
        z, x_root = ctx.saved_tensors
        gy = gy.contiguous()
        gw = gb = gwr = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:
                gw, gb = gemm_tn(gy, z, colsum=True)
            else:
                gw = gemm_tn(gy, z)
        elif want_b:
            gb = gy.sum(0)
        if x_root is not None and ctx.needs_input_grad[3]:
            gwr = gemm_tn(gy, x_root)
        return None, gw, gb, gwr, None, None


def aggregate_linear_ok(in_channels, out_channels, root=False):
    return bool(_lib.load().rgbx_spmm_linear_supported(in_channels, out_channels, int(root)))


def aggregate_linear(z, weight, bias=None, root_weight=None, x_root=None, want_colsums=False):
    """Transform of a kept aggregate (see _AggregateLinear); the caller checked aggregate_linear_ok."""
    return _tag_colsums(_AggregateLinear.apply(z, weight, bias, root_weight, x_root, want_colsums), want_colsums)


def fold_bn_linear(weight, bias=None, bias2=None, root_weight=None, bn=None):
    """(W'^T [K, Nout], b' [Nout] or None, Wr'^T or None): the operands rgbx_fused_layer_f32 reads for a linear layer
    with the eval-mode BatchNorm1d `bn` behind it folded in (bn None: the plain transposes), in ONE launch
    (rgbx_fold_bn_linear_f32). No autograd: eval forwards only."""
    _lib.require_device(weight, bias, bias2, root_weight)
    W = weight.detach()
    W = W if W.stride(-1) == 1 else W.contiguous()
    n_out, K = W.shape
    Wr = None
    if root_weight is not None:
        Wr = root_weight.detach()
        Wr = Wr if Wr.stride(-1) == 1 else Wr.contiguous()
    wt = torch.empty((K, n_out), dtype=torch.float32, device=W.device)
    wrt = torch.empty_like(wt) if Wr is not None else None
    has_b = bias is not None or bias2 is not None or bn is not None
    b_out = torch.empty(n_out, dtype=torch.float32, device=W.device) if has_b else None
    c = lambda t: None if t is None else t.detach().contiguous()
    b1, b2 = c(bias), c(bias2)
    g = be = rm = rv = None
    eps = 0.0
    if bn is not None:
        g, be, rm, rv, eps = c(bn.weight), c(bn.bias), c(bn.running_mean), c(bn.running_var), bn.eps
    _lib.check(_lib.load().rgbx_fold_bn_linear_f32(
        _lib.ptr(W), W.stride(0), _lib.ptr(Wr), 0 if Wr is None else Wr.stride(0), _lib.ptr(b1), _lib.ptr(b2),
        _lib.ptr(g), _lib.ptr(be), _lib.ptr(rm), _lib.ptr(rv), float(eps), _lib.ptr(wt), _lib.ptr(wrt), _lib.ptr(b_out),
        n_out, K, _lib.stream_ptr()), "rgbx_fold_bn_linear_f32")
    return wt, b_out, wrt


class _PropagateLinear(torch.autograd.Function):
    """y = (P x) W^T + b (+ x Wr^T) with P = A_hat ('gcn'), the mean operator ('mean') or the plain edge sum ('sum'), in one launch
    (rgbx_spmm_linear_f32). Backward: dW = dy^T (P x) on the split-K MFMA kernel (P x was stored by the
    forward when a gradient is needed), db = column sums from the same pass, dWr = dy^T x, and — only if x
    needs a gradient — dx = P^T (dy W) + dy Wr = (P^T dy) W + dy Wr: the SAME fused kernel on the transposed CSR
    (W as stored is already the [K, Nout] operand) when in == out, else GEMMs and the transposed SpMM with the
    root part as its additive term."""

    @staticmethod
    def forward(ctx, x, graph, kind, weight, bias, need_z=True, root_weight=None, x_root=None, want_colsums=False):
        """`x_root`: the targets' own rows when they are not simply the first rows of `x` viewed as a separate
        tensor (partitioned graph: x = [local; halo], x_root = local). `want_colsums`: returns (out, colsums)."""
        x = x.contiguous()
        xr = x if x_root is None else x_root.contiguous()
        w = graph.w if kind == "gcn" else None          # properties: each builds its vector on first use only
        rs = graph.inv_deg if kind == "mean" else None  # ('sum': neither)
        if kind not in ("gcn", "mean", "sum"):
            raise ValueError(kind)
        res = spmm_linear_raw(graph.fwd, w, rs, x, weight_t(weight), None if bias is None else bias.detach(),
                              need_z, xr if root_weight is not None else None,
                              None if root_weight is None else weight_t(root_weight),
                              kind=f"{getattr(graph, 'event_prefix', '')}{kind}_linear_fwd", want_colsums=want_colsums)
        out, z = res[0], res[1]
        ctx.save_for_backward(z, weight, root_weight, xr if root_weight is not None else None)
        ctx.graph, ctx.kind, ctx.has_bias = graph, kind, bias is not None
        if want_colsums:
            ctx.mark_non_differentiable(res[2])
            return out, res[2]
        return out

    @staticmethod
    def backward(ctx, gy, _g_colsums=None):
        z, weight, root_weight, x = ctx.saved_tensors
        g, kind = ctx.graph, ctx.kind
        gy = gy.contiguous()
        gx = gw = gb = gwr = None
        want_b = ctx.has_bias and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[3]:
            if want_b:
                gw, gb = gemm_tn(gy, z, colsum=True)  # dy is read once for dW and db
            else:
                gw = gemm_tn(gy, z)
        elif want_b:
            gb = gy.sum(0)
        if root_weight is not None and ctx.needs_input_grad[6]:
            gwr = gemm_tn(gy, x)
        if ctx.needs_input_grad[0]:
            gx = _propagate_linear_input_grad(g, kind, gy, weight, root_weight)
        return gx, None, None, gw, gb, None, gwr, None, None


def _propagate_linear_input_grad(g, kind, gy, weight, root_weight):
    """dx = P^T (dy W) + dy Wr = (P^T dy) W + dy Wr: the fused kernel on the transposed CSR (W as stored is already
    the [K, Nout] operand) when in == out, else GEMMs and the transposed SpMM with the root part as its additive
    term."""
    wt = {"gcn": lambda: g.w_t, "mean": lambda: g.w_mean_t, "sum": lambda: None}[kind]()
    prefix = getattr(g, "event_prefix", "")
    n_out, n_in = weight.shape
    if n_out <= n_in and _lib.load().rgbx_spmm_linear_supported(n_out, n_in, int(root_weight is not None)):
        gx, _ = spmm_linear_raw(g.bwd, wt, None, gy, weight.detach().contiguous(), None, False,
                                gy if root_weight is not None else None,
                                None if root_weight is None else root_weight.detach().contiguous(),
                                kind=f"{prefix}{kind}_linear_bwd")
        return gx
    gz = gy @ weight
    gr = gy @ root_weight if root_weight is not None else None
    if gr is None:
        return spmm_raw(g.bwd, wt, None, gz, kind=f"{prefix}{kind}_bwd")
    return spmm_raw(g.bwd, wt, None, gz, y=gr, a=1.0, b=1.0, out=gr, kind=f"{prefix}{kind}_bwd")


class _BNPropagateLinear(torch.autograd.Function):
    """y = (P BN(x)) W^T + b (+ BN(x) Wr^T) for a TRAINING-mode BatchNorm1d in front of a conv layer
    (models/gcn.py:28-29: x = bns[i](x); x = convs[i+1](x, edge_index)) without writing BN(x): the statistics pass
    gives BN(x) = x * s + t per column, and rgbx_spmm_linear_f32 gathers the RAW rows and maps the aggregate,
    s * (P x) + t * rowsum(P) (linearity of the aggregation), before the MFMA transform.
    Backward = the two layers' own backward passes one after the other, unchanged: dW = dy^T z (z = the mapped
    aggregate the forward stored), the gradient g_h of the BatchNorm output from the fused kernel on the transposed
    CSR, then BatchNorm's backward (column sums of g_h and g_h * xhat, apply)."""

    @staticmethod
    def forward(ctx, x, bn_weight, bn_bias, graph, kind, weight, bias, root_weight, eps, reduce, running, need_z,
                colsums=None, want_colsums=False):
        from .nn import batchnorm as B
        x = x.contiguous()
        mean, rstd, scale, shift, n = B.train_statistics(x, bn_weight, bn_bias, eps, reduce, running, colsums)
        w, rs = (graph.w, None) if kind == "gcn" else (None, graph.inv_deg)
        res = spmm_linear_raw(graph.fwd, w, rs, x, weight_t(weight), None if bias is None else bias.detach(), need_z,
                              x if root_weight is not None else None,
                              None if root_weight is None else weight_t(root_weight), kind=f"{kind}_linear_fwd",
                              pre=(scale, shift, graph.rowsum(kind)), want_colsums=want_colsums)
        out, z = res[0], res[1]
        ctx.save_for_backward(x, bn_weight, mean, rstd, n, z, weight, root_weight, scale, shift)
        ctx.graph, ctx.kind, ctx.has_bias, ctx.reduce = graph, kind, bias is not None, reduce
        if want_colsums:
            ctx.mark_non_differentiable(res[2])
            return out, res[2]
        return out

    @staticmethod
    def backward(ctx, gy, _g_colsums=None):
        from .nn import batchnorm as B
        x, bn_weight, mean, rstd, n, z, weight, root_weight, scale, shift = ctx.saved_tensors
        g, kind = ctx.graph, ctx.kind
        gy = gy.contiguous()
        gw = gb = gwr = None
        if ctx.needs_input_grad[5]:
            gw, gcol = gemm_tn(gy, z, colsum=True)  # dy is read once for dW and the column sums
            gb = gcol if ctx.has_bias and ctx.needs_input_grad[6] else None
        else:
            gcol = gy.sum(0)
            gb = gcol if ctx.has_bias and ctx.needs_input_grad[6] else None
        if root_weight is not None and ctx.needs_input_grad[7]:
            # dWr = dy^T BN(x) = (dy^T x) diag(s) + colsum(dy) t^T
            gwr = gemm_tn(gy, x) * scale + gcol[:, None] * shift
        g_h = _propagate_linear_input_grad(g, kind, gy, weight, root_weight)
        gx, g_bnw, g_bnb = B.train_backward(g_h, x, bn_weight, mean, rstd, n, ctx.reduce)
        return gx, g_bnw, g_bnb, None, None, gw, gb, gwr, None, None, None, None, None, None


_SCALE_CACHE = {}


def mask_scale(y, mask, C):
    """Device scalar 1 / (number of rows the masked cross-entropy selects): rows with mask set (all if None) and a
    label in [0, C). A mask is fixed for a run, so the count is taken once per (labels, mask) pair — the fused loss
    epilogue needs the mean's divisor BEFORE the launch that produces the logits."""
    key = (y.data_ptr(), y._version, y.numel(), None if mask is None else (mask.data_ptr(), mask._version), int(C))
    hit = _SCALE_CACHE.get(key)
    if hit is None:
        sel = (y >= 0) & (y < C)
        if mask is not None:
            sel = sel & mask.bool()
        hit = (_MadeOn((1.0 / sel.sum().double()).float().reshape(1)), y, mask)  # the tensors stay alive with their addresses
        _SCALE_CACHE[key] = hit
        while len(_SCALE_CACHE) > 32:
            _SCALE_CACHE.pop(next(iter(_SCALE_CACHE)))
    return hit[0].get()


_GROUP_MASKS = {}


def group_masks(mask_a, mask_b):
    """uint8 [N] with bit 0 = mask_a, bit 1 = mask_b (rgbx_ce_epilogue_t.mask_groups == 2): made once per pair of mask
    tensors (address + version), masks are fixed for a run."""
    _lib.require_device(mask_a, mask_b)
    key = (mask_a.data_ptr(), mask_a._version, mask_b.data_ptr(), mask_b._version, mask_a.numel())
    hit = _GROUP_MASKS.get(key)
    if hit is None:
        g = (mask_a.bool().to(torch.uint8) | (mask_b.bool().to(torch.uint8) << 1)).contiguous()
        hit = (_MadeOn(g), mask_a, mask_b)  # the mask tensors stay alive with their addresses
        _GROUP_MASKS[key] = hit
        while len(_GROUP_MASKS) > 16:
            _GROUP_MASKS.pop(next(iter(_GROUP_MASKS)))
    return hit[0].get()


def ce_from_logits(logits, y, mask):
    """(loss, stats) of the masked cross-entropy taken from materialised logits: the unfused route of the *_ce
    entry points. loss = NLLLoss(log_softmax(logits)[mask], y[mask]); stats = [nll sum, selected rows, correct].
    `mask` = (mask_a, mask_b): (None, [2, 3] stats) of both masks from the one set of logits."""
    if isinstance(mask, (tuple, list)):
        return None, torch.stack([masked_ce_accuracy(logits, y, m) for m in mask])
    if torch.is_grad_enabled() and logits.requires_grad:
        return masked_ce_loss(logits, y, mask, with_stats=True)
    stats = masked_ce_accuracy(logits, y, mask)
    return (stats[0] / stats[1]).float(), stats


def fused_ce_ok(graph, in_channels, out_channels, root, x, y):
    """The model's last conv can take the loss into its kernel: fused aggregate+transform applies, at most 128
    classes, single-GPU graph, device labels."""
    return (not _is_dist(graph) and out_channels <= 128 and y is not None and y.is_cuda and y.dtype == torch.int64
            and fused_linear_ok(graph, in_channels, out_channels, root=root, x=x))


class _PropagateLinearCE(torch.autograd.Function):
    """loss = masked cross-entropy of ((P x') W^T + b (+ x' Wr^T)) with x' = BN(x) (training BatchNorm handed over as
    in _BNPropagateLinear) or x itself — the model's last conv with the loss taken inside the kernel
    (rgbx_ce_epilogue_t): the logits are never written. A forward that prepares a backward stores the loss gradient
    w.r.t. the logits instead (scaled by 1 / selected rows); the backward is then _PropagateLinear's / _BNPropagateLinear's
    with that matrix as dy and the incoming scalar folded into the small operands (W, Wr, dW, db)."""

    @staticmethod
    def forward(ctx, x, graph, kind, weight, bias, root_weight, y, mask, want_grad, bn_weight, bn_bias, bn_args):
        from .nn import batchnorm as B
        x = x.contiguous()
        pre = None
        if bn_weight is not None:
            eps, reduce, running, colsums = bn_args
            mean, rstd, scale, shift, n = B.train_statistics(x, bn_weight, bn_bias, eps, reduce, running, colsums)
            pre = (scale, shift, graph.rowsum(kind))
        w, rs = (graph.w, None) if kind == "gcn" else (None, graph.inv_deg)
        grad_scale = mask_scale(y, mask, weight.size(0)) if want_grad else None
        dlogits, z, stats = spmm_linear_raw(
            graph.fwd, w, rs, x, weight_t(weight), None if bias is None else bias.detach(),
            want_grad and weight.requires_grad, x if root_weight is not None else None,
            None if root_weight is None else weight_t(root_weight), kind=f"{kind}_linear_fwd", pre=pre,
            ce=(y, mask, grad_scale))
        ctx.graph, ctx.kind, ctx.has_bias, ctx.has_bn = graph, kind, bias is not None, bn_weight is not None
        if want_grad:
            if ctx.has_bn:
                ctx.save_for_backward(dlogits, z, weight, root_weight, x, bn_weight, mean, rstd, n, scale, shift)
                ctx.reduce = bn_args[1]
            else:
                ctx.save_for_backward(dlogits, z, weight, root_weight, x if root_weight is not None else None)
        ctx.mark_non_differentiable(stats)
        if stats.dim() == 2:  # two masks (statistics only): the caller reads the [2, 3] table
            return stats.new_zeros(()).float(), stats
        return (stats[0] / stats[1]).float(), stats

    @staticmethod
    def backward(ctx, g, _g_stats):
        from .nn import batchnorm as B
        saved = ctx.saved_tensors
        if ctx.has_bn:
            gy, z, weight, root_weight, x, bn_weight, mean, rstd, n, scale, shift = saved
        else:
            gy, z, weight, root_weight, x = saved
        graph, kind = ctx.graph, ctx.kind
        g = g.reshape(()).float()
        gw = gb = gwr = gx = g_bnw = g_bnb = None
        gcol = None
        if ctx.needs_input_grad[3]:  # z was stored by the forward exactly when the weight wants a gradient
            gw, gcol = gemm_tn(gy, z, colsum=True)
        if ctx.has_bias and ctx.needs_input_grad[4]:
            gb = gcol if gcol is not None else gy.sum(0)
        if root_weight is not None and ctx.needs_input_grad[5]:
            if ctx.has_bn:  # dWr = dy^T BN(x) = (dy^T x) diag(s) + colsum(dy) t^T
                gcol = gcol if gcol is not None else gy.sum(0)
                gwr = gemm_tn(gy, x) * scale + gcol[:, None] * shift
            else:
                gwr = gemm_tn(gy, x)
        need_h = ctx.needs_input_grad[0] or (ctx.has_bn and (ctx.needs_input_grad[9] or ctx.needs_input_grad[10]))
        # the incoming scalar multiplies everything linear in dy: the small operands take it, in ONE multi-tensor launch
        items = [(k, t) for k, t in (("gw", gw), ("gb", gb), ("gwr", gwr)) if t is not None]
        if need_h:
            items.append(("w", weight.detach()))
            if root_weight is not None:
                items.append(("wr", root_weight.detach()))
        sc = dict(zip((k for k, _ in items), torch._foreach_mul([t for _, t in items], g))) if items else {}
        gw, gb, gwr = sc.get("gw"), sc.get("gb"), sc.get("gwr")
        if need_h:
            g_h = _propagate_linear_input_grad(graph, kind, gy, sc["w"], sc.get("wr"))
            if ctx.has_bn:
                gx, g_bnw, g_bnb = B.train_backward(g_h, x, bn_weight, mean, rstd, n, ctx.reduce)
            else:
                gx = g_h
        return gx, None, None, gw, gb, gwr, None, None, None, g_bnw, g_bnb, None


def propagate_linear_ce(x, graph, kind, weight, bias, root_weight, y, mask, bn=None, colsums=None):
    """(loss, stats) = masked cross-entropy of the last conv's logits, taken inside rgbx_spmm_linear_f32 (the
    caller checked fused_ce_ok, and bn.folds_into_next_layer when a training BatchNorm `bn` is handed over)."""
    # any operand a gradient can reach (frozen-weight fine-tuning: only a bias or the root weight may want one)
    want_grad = not isinstance(mask, (tuple, list)) and torch.is_grad_enabled() and any(
        t is not None and t.requires_grad
        for t in (weight, x, bias, root_weight) + ((bn.weight, bn.bias) if bn is not None else ()))
    if bn is not None:
        bn_args = (bn.eps, bn._reduce, bn.begin_training_step(), colsums)
        return _PropagateLinearCE.apply(x, graph, kind, weight, bias, root_weight, y, mask, want_grad, bn.weight, bn.bias,
                                        bn_args)
    return _PropagateLinearCE.apply(x, graph, kind, weight, bias, root_weight, y, mask, want_grad, None, None, None)


def bn_propagate_linear(x, bn, graph, kind, weight, bias=None, root_weight=None, colsums=None, want_colsums=False):
    """conv(bn(x)) for a training-mode BatchNorm1d `bn` and a conv layer whose propagate runs on the fused
    aggregate+transform kernel (single-GPU graphs; the caller checked fused_linear_ok and bn.folds_into_next_layer).
    `colsums`: the [2, d] column sums of x and x^2 if the launch that produced x already took them."""
    need_z = weight.requires_grad
    return _tag_colsums(_BNPropagateLinear.apply(x, bn.weight, bn.bias, graph, kind, weight, bias, root_weight, bn.eps,
                                                 bn._reduce, bn.begin_training_step(), need_z, colsums, want_colsums),
                        want_colsums)


COLSUMS = "_rgbx_colsums"  # attribute a conv output carries when the launch also produced its column sums


def _tag_colsums(res, want_colsums):
    """(out, colsums) of a Function -> out, with the [2, Nout] float64 column sums of out and out^2 riding on it as
    an attribute for the BatchNorm that follows (models/_stack.ConvStack reads it right after the conv returns)."""
    if not want_colsums:
        return res
    out, colsums = res
    setattr(out, COLSUMS, colsums)
    return out


def propagate_linear(x, graph, kind, weight, bias=None, root_weight=None, want_colsums=False):
    if _is_dist(graph):  # resident input features of a partitioned graph (fused_linear_ok checked it)
        return graph.propagate_linear(x, kind, weight, bias, root_weight)
    # the aggregate is kept only when the weight gradient (dy^T (P x)) will be asked for; Function.forward
    # cannot see the caller's grad mode, so the decision is taken here
    need_z = torch.is_grad_enabled() and weight.requires_grad
    return _tag_colsums(_PropagateLinear.apply(x, graph, kind, weight, bias, need_z, root_weight, None, want_colsums),
                        want_colsums)


def appnp_raw(csr, w, h, K, alpha, kind="appnp"):
    _lib.require_device(h)
    h = h.contiguous()
    ph, ldh = _lib.mat(h, "h")
    out = torch.empty_like(h)
    tmp = torch.empty_like(h) if K > 1 else None
    po, ldo = _lib.mat(out, "out")
    split, _scratch = csr.split_arg(h.size(1), h.device)
    with _Timed(kind):
        _lib.check(
            _lib.load().rgbx_appnp_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(w), ph, ldh, po,
                                       _lib.ptr(tmp), ldo, csr.N, h.size(1), int(K), float(alpha),
                                       None if split is None else ctypes.byref(split), _lib.stream_ptr()),
            "rgbx_appnp_f32")
    return out


class _APPNP(torch.autograd.Function):
    """z <- (1-alpha) A_hat z + alpha h, K times (appnp_stack.py:29; twin pta.py:79-84).

    Backward is the same recurrence on the transposed graph applied to the incoming gradient:
    dL/dh = M^K g + alpha * sum_{j<K} M^j g with M = (1-alpha) A_hat^T (Horner form)."""

    @staticmethod
    def forward(ctx, h, graph, K, alpha):
        ctx.graph, ctx.K, ctx.alpha = graph, K, alpha
        return appnp_raw(graph.fwd, graph.w, h, K, alpha, kind="appnp_fwd")

    @staticmethod
    def backward(ctx, gy):
        g = ctx.graph
        return appnp_raw(g.bwd, g.w_t, gy, ctx.K, ctx.alpha, kind="appnp_bwd"), None, None, None


class _APPNPCE(torch.autograd.Function):
    """(loss, stats) of the masked cross-entropy of APPNP's output (models/appnp_stack.py:29-31 followed by the loss of
    itexperiments.py:429 / the metrics of :624-626): steps 1..K-1 as rgbx_appnp_f32, the last step on the row kernel with
    the loss epilogue — the [N, C] logits are never written. `h` is [N, n] with columns [n_classes, n) zero padding.
    Backward: the recurrence on the transposed graph applied to the stored loss gradient (see _APPNP)."""

    @staticmethod
    def forward(ctx, h, graph, K, alpha, labels, mask, n_classes, want_grad):
        h = h.contiguous()
        z = appnp_raw(graph.fwd, graph.w, h, K - 1, alpha, kind="appnp_fwd") if K > 1 else h
        grad_scale = mask_scale(labels, mask, n_classes) if want_grad else None
        dlogits, stats = spmm_epilogue_raw(graph.fwd, graph.w, None, z, y=h, a=1.0 - alpha, b=alpha,
                                           ce=(labels, mask, grad_scale), n_classes=n_classes, kind="appnp_fwd")
        ctx.graph, ctx.K, ctx.alpha = graph, K, alpha
        if want_grad:
            ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(stats)
        if stats.dim() == 2:
            return stats.new_zeros(()).float(), stats
        return (stats[0] / stats[1]).float(), stats

    @staticmethod
    def backward(ctx, g, _g_stats=None):
        if not ctx.saved_tensors:
            raise RuntimeError("appnp_propagate_ce: backward asked of a forward that was run without want_grad")
        gy = ctx.saved_tensors[0] * g.reshape(()).float()
        gr = ctx.graph
        return appnp_raw(gr.bwd, gr.w_t, gy, ctx.K, ctx.alpha, kind="appnp_bwd"), None, None, None, None, None, None, None


def appnp_propagate_ce(h, graph, K, alpha, n_classes, y, mask):
    """(loss, stats) of APPNP(K, alpha)(h)'s masked cross-entropy with the loss inside the last step's kernel; h [N, n],
    n % 4 == 0, columns beyond n_classes zero (pad_rows4). The caller checked rows_epilogue_ok and K >= 1."""
    pair = isinstance(mask, (tuple, list))
    want_grad = (not pair) and torch.is_grad_enabled() and h.requires_grad
    loss, stats = _APPNPCE.apply(h, graph, int(K), float(alpha), y, mask, int(n_classes), want_grad)
    return (None if pair else loss), stats


def appnp_propagate(h, graph, K, alpha):
    if _is_dist(graph):
        out = graph.appnp(h, K, alpha)  # reshard scheme: all K steps inside one pair of transposes
        if out is not None:
            return out
        z = h  # halo scheme: one exchange per iteration; autograd chains the K distributed propagates
        for _ in range(K):
            z = (1.0 - alpha) * graph.propagate(z, "gcn") + alpha * h
        return z
    hp, d = _pad4(h)  # C = 7 classes: all K propagates on 8-float rows
    out = _APPNP.apply(hp, graph, K, alpha)
    return out if hp is h else out[:, :d]


class _DAGNNProp(torch.autograd.Function):
    """Prop.forward of DAGNN (reference models/dagnn.py:41-55): K gcn-normalised propagates, then
    out = sum_k sigmoid(<hop_k, s> + b) * hop_k over the K+1 hops. The hops are written by the SpMM into one
    [K, N, d] buffer and read once by rgbx_dagnn_gate_fwd_f32 — no stack, no [N, K+1, d] projection input.
    Backward: one pass (rgbx_dagnn_gate_bwd_f32) writes every hop's direct gradient and reduces g_s / g_b; the
    chain through the propagates is the Horner form G_k = A_hat^T G_{k+1} + direct_k, K transposed SpMMs that add
    their `y` operand in the store."""

    @staticmethod
    def forward(ctx, x, graph, K, s, b):
        lib = _lib.load()
        N, d = x.shape
        x = x.contiguous()
        hops = torch.empty((K, N, d), dtype=torch.float32, device=x.device)
        src = x
        for k in range(K):
            spmm_raw(graph.fwd, graph.w, None, src, out=hops[k], kind="gcn_fwd")
            src = hops[k]
        out = torch.empty_like(x)
        sv = s.detach().reshape(-1).contiguous()
        bv = None if b is None else b.detach().reshape(-1).contiguous()
        with _Timed("dagnn_gate_fwd"):
            _lib.check(lib.rgbx_dagnn_gate_fwd_f32(_lib.ptr(x), x.stride(0), _lib.ptr(hops), N * d, d, _lib.ptr(sv),
                                                   _lib.ptr(bv), _lib.ptr(out), out.stride(0), N, d, K,
                                                   _lib.stream_ptr()), "rgbx_dagnn_gate_fwd_f32")
        ctx.graph, ctx.K, ctx.has_b = graph, K, b is not None
        ctx.save_for_backward(x, hops, sv, bv)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        x, hops, sv, bv = ctx.saved_tensors
        g, K = ctx.graph, ctx.K
        N, d = x.shape
        gout = gout.contiguous()
        d0 = torch.empty_like(x)
        dk = torch.empty((K, N, d), dtype=torch.float32, device=x.device)
        g_s = torch.empty(d, dtype=torch.float32, device=x.device)
        g_b = torch.empty(1, dtype=torch.float32, device=x.device) if ctx.has_b else None
        need = ctypes.c_size_t(0)
        _lib.check(lib.rgbx_dagnn_gate_bwd_workspace_bytes(d, ctypes.byref(need)), "rgbx_dagnn_gate_bwd_workspace_bytes")
        ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
        with _Timed("dagnn_gate_bwd"):
            _lib.check(lib.rgbx_dagnn_gate_bwd_f32(_lib.ptr(x), x.stride(0), _lib.ptr(hops), N * d, d, _lib.ptr(sv),
                                                   _lib.ptr(bv), _lib.ptr(gout), gout.stride(0), _lib.ptr(d0),
                                                   d0.stride(0), _lib.ptr(dk), N * d, d, _lib.ptr(g_s), _lib.ptr(g_b),
                                                   _lib.ptr(ws), need.value, N, d, K, _lib.stream_ptr()),
                       "rgbx_dagnn_gate_bwd_f32")
        gx = None
        if ctx.needs_input_grad[0]:
            # G_K = direct_K; G_k = A_hat^T G_{k+1} + direct_k, each written over direct_k
            cur = dk[K - 1] if K else d0
            for k in range(K - 1, -1, -1):
                dst = dk[k - 1] if k else d0
                spmm_raw(g.bwd, g.w_t, None, cur, y=dst, a=1.0, b=1.0, out=dst, kind="gcn_bwd")
                cur = dst
            gx = d0
        return gx, None, None, g_s, g_b


def dagnn_prop(x, graph, K, proj_weight, proj_bias=None):
    """DAGNN's propagate-and-mix (reference models/dagnn.py:41-55) on the HIP path; proj_weight [1, C], proj_bias [1].
    Returns None when this shape is left to the caller's torch formulation (a partitioned graph, C > 256)."""
    if _is_dist(graph) or x.size(1) > 256:
        return None
    _lib.require_device(x, proj_weight, proj_bias)
    xp, d = _pad4(x)
    s = proj_weight.reshape(-1)
    if xp is not x:
        s = torch.nn.functional.pad(s, (0, xp.size(1) - d))  # zero columns: no part in the scores or the mix
    out = _DAGNNProp.apply(xp, graph, int(K), s, proj_bias)
    return out if xp is x else out[:, :d]


class _GATScores(torch.autograd.Function):
    """a_src[n,h] = <hfeat[n,h,:], att_src[h,:]> and a_dst likewise (GATConv.forward [PyG])."""

    @staticmethod
    def forward(ctx, hfeat, att_src, att_dst, H, C):
        _lib.require_device(hfeat, att_src, att_dst)
        hfeat = hfeat.contiguous()
        n = hfeat.size(0)
        a_src = torch.empty((n, H), dtype=torch.float32, device=hfeat.device)
        a_dst = torch.empty_like(a_src)
        ph, ldh = _lib.mat(hfeat, "hfeat")
        att_s = att_src.reshape(H, C).contiguous()
        att_d = att_dst.reshape(H, C).contiguous()
        _lib.check(
            _lib.load().rgbx_gat_scores_f32(ph, ldh, _lib.ptr(att_s), _lib.ptr(att_d), _lib.ptr(a_src),
                                            _lib.ptr(a_dst), n, H, C, _lib.stream_ptr()), "rgbx_gat_scores_f32")
        ctx.save_for_backward(hfeat, att_s, att_d)
        ctx.H, ctx.C, ctx.att_shape = H, C, att_src.shape
        return a_src, a_dst

    @staticmethod
    def backward(ctx, g_as, g_ad):
        hfeat, att_s, att_d = ctx.saved_tensors
        H, C = ctx.H, ctx.C
        h3 = hfeat.view(-1, H, C)
        g_h = (g_as.unsqueeze(-1) * att_s + g_ad.unsqueeze(-1) * att_d).reshape(hfeat.shape)
        g_att_s = torch.einsum("nh,nhc->hc", g_as, h3).reshape(ctx.att_shape)
        g_att_d = torch.einsum("nh,nhc->hc", g_ad, h3).reshape(ctx.att_shape)
        return g_h, g_att_s, g_att_d, None, None


class _GATAggregate(torch.autograd.Function):
    """out[i,h,:] = sum_j softmax_j(leaky_relu(a_src[j,h] + a_dst[i,h])) * hfeat[j,h,:]."""

    @staticmethod
    def forward(ctx, hfeat, a_src, a_dst, graph, H, C, slope, att_src=None, want_grad=True):
        """`att_src` ([H, C], no gradient through this argument: the score path's gradient flows through
        `a_src`) lets the kernel form the source scores from the rows it gathers anyway. `want_grad`: a backward
        will follow (the forward then also stores the positive-score parts the backward's prep pass needs)."""
        _lib.require_device(hfeat, a_src, a_dst, att_src)
        hfeat, a_src, a_dst = hfeat.contiguous(), a_src.contiguous(), a_dst.contiguous()
        att = None if att_src is None else att_src.detach().reshape(H, C).contiguous()
        N = graph.fwd.N  # targets; hfeat / a_src may have more rows (sources incl. a halo) than targets
        dev = hfeat.device
        out = torch.empty((N, H * C), dtype=torch.float32, device=dev)
        m = torch.empty((N, H), dtype=torch.float32, device=dev)
        rden = torch.empty_like(m)
        ph, ldh = _lib.mat(hfeat, "hfeat")
        po, ldo = _lib.mat(out, "out")
        csr = graph.fwd
        opos, apos = _gat_train_extras(want_grad, N, H, C, dev)
        split, _scratch = csr.split_arg((2 * H * C + 3 * H) if want_grad else (H * C + 2 * H), dev)
        with _Timed("gat_fwd"):
            _lib.check(
                _lib.load().rgbx_gat_aggregate_fwd_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), ph, ldh,
                                                       _lib.ptr(a_src), _lib.ptr(att), _lib.ptr(a_dst), None, None, None,
                                                       po, ldo, _lib.ptr(m), _lib.ptr(rden), _lib.ptr(opos),
                                                       _lib.ptr(apos), N, H, C, float(slope),
                                                       None if split is None else ctypes.byref(split),
                                                       _lib.stream_ptr()), "rgbx_gat_aggregate_fwd_f32")
        ctx.save_for_backward(hfeat, a_src, a_dst, m, rden, out, opos, apos)
        ctx.graph, ctx.H, ctx.C, ctx.slope = graph, H, C, slope
        return out

    @staticmethod
    def backward(ctx, gout):
        """Two launches, one full gather pass: (1) per-target records nodeq = (a_dst, max - log(1/sum),
        <gout, out>) and the target-side score gradient g_a_dst in a streaming pass; (2) the source-side pass
        over the transposed CSR gathers gout rows + records and produces g_hfeat and g_a_src."""
        hfeat, a_src, a_dst, m, rden, out, opos, apos = ctx.saved_tensors
        g_h, g_as, g_ad = _gat_backward_core(ctx.graph, hfeat, a_src, a_dst, m, rden, out, gout.contiguous(), ctx.H,
                                             ctx.C, ctx.slope, opos=opos, apos=apos)
        return g_h, g_as, g_ad, None, None, None, None, None, None


def _gat_train_extras(want_grad, N, H, C, dev):
    """(out_pos [N, H*C], a_pos [N, H]) buffers of a forward whose backward will be asked for, else (None, None)."""
    if not want_grad:
        return None, None
    return (torch.empty((N, H * C), dtype=torch.float32, device=dev),
            torch.empty((N, H), dtype=torch.float32, device=dev))


def _gat_backward_core(g, hfeat, a_src, a_dst, m, rden, out, gout, H, C, slope, bias=None, opos=None, apos=None,
                       att=None):
    """(g_hfeat through the aggregation, g_a_src [n_src, H], g_a_dst [n_tgt, H]). `bias`: the vector the
    forward added to `out` in its store, if any. `opos` / `apos`: the positive-score parts the forward stored
    (rgbx_gat_aggregate_fwd_f32 out_pos / a_pos): with them g_a_dst comes out of the streaming prep pass; without
    them (a forward run without want_grad) the source-side pass also writes the per-edge score gradient ds [E', H]
    and g_a_dst is its width-H segment sum over each target's in-edges. `att` = (att_src, att_dst) [H, C] (needs
    opos / apos): the scores are products of hfeat with these vectors, so the source pass folds their backward into
    its g_hfeat store (and forms the source's own score from its row when `a_src` is None)."""
    N, n_src, dev = g.fwd.N, g.bwd.N, gout.device
    lib = _lib.load()
    nodeq = torch.empty((N, H, 4), dtype=torch.float32, device=dev)  # (a_dst, max - log(1/sum), dsum, -) records
    g_as = torch.empty((n_src, H), dtype=torch.float32, device=dev)
    g_h = torch.empty_like(hfeat)
    ph, ldh = _lib.mat(hfeat, "hfeat")
    po, ldo = _lib.mat(out, "out")
    pg, ldg = _lib.mat(gout, "gout")
    pgh, ldgh = _lib.mat(g_h, "g_hfeat")
    per_node = opos is not None
    fold = per_node and att is not None
    if a_src is None and not fold:
        raise RuntimeError("gat backward: a_src may only be left to the kernel together with the folded score backward")
    g_ad = None
    if per_node:  # the fold reads g_a_dst by SOURCE row: rows that are no target (halo rows of a partitioned run) are 0
        g_ad_full = (torch.zeros if fold and n_src > N else torch.empty)((max(n_src, N) if fold else N, H),
                                                                         dtype=torch.float32, device=dev)
        g_ad = g_ad_full[:N]
    att2 = torch.cat([att[0].reshape(1, H, C), att[1].reshape(1, H, C)]).contiguous() if fold else None
    ds = None if per_node else torch.empty((max(g.bwd.nnz, 1), H), dtype=torch.float32, device=dev)
    with _Timed("gat_bwd_prep"):
        _lib.check(
            lib.rgbx_gat_bwd_prep_f32(_lib.ptr(a_dst), _lib.ptr(m), _lib.ptr(rden), po, ldo, _lib.ptr(bias), pg, ldg,
                                      _lib.ptr(nodeq), _lib.ptr(opos), _lib.ptr(apos), float(slope), _lib.ptr(g_ad),
                                      N, H, C, _lib.stream_ptr()), "rgbx_gat_bwd_prep_f32")
    split, _scratch = g.bwd.split_arg(H * C + 2 * H, dev)
    with _Timed("gat_bwd_src"):
        _lib.check(
            lib.rgbx_gat_bwd_src_f32(_lib.ptr(g.bwd.rowptr), _lib.ptr(g.bwd.col), ph, ldh, _lib.ptr(a_src),
                                     _lib.ptr(nodeq), pg, ldg, pgh, ldgh, _lib.ptr(g_as), _lib.ptr(ds),
                                     _lib.ptr(att2), _lib.ptr(g_ad_full) if fold else None, n_src, H, C, float(slope),
                                     None if split is None else ctypes.byref(split), _lib.stream_ptr()),
            "rgbx_gat_bwd_src_f32")
    if not per_node:
        g_ad = spmm_raw(gat_segment_csr(g), None, None, ds, kind="gat_bwd_segsum")
    return g_h, g_as, g_ad


def gat_segment_csr(graph):
    """CSR over the targets whose `col` is, per forward slot, the TRANSPOSED slot of the same edge: summing
    rows of the per-edge tensor `ds` (stored in transposed-slot order) through it gives per-target totals.
    Built once per graph from the two slot -> edge-id permutations."""
    seg = getattr(graph, "_gat_seg", None)
    if seg is None:
        from .graph import CSR
        f, b = graph.fwd, graph.bwd
        if f.nnz:
            inv = torch.empty(int(max(f.perm[:f.nnz].max(), b.perm[:b.nnz].max()).item()) + 1, dtype=torch.int32,
                              device=f.perm.device)
            inv[b.perm[:b.nnz].long()] = torch.arange(b.nnz, dtype=torch.int32, device=inv.device)
            col = inv[f.perm[:f.nnz].long()].contiguous()
        else:
            col = f.col
        seg = CSR(f.rowptr, col, f.perm, f.N, f.nnz, f.split)
        graph._gat_seg = seg
    return seg


def gat_scores(hfeat, att_src, att_dst, H, C):
    return _GATScores.apply(hfeat, att_src, att_dst, H, C)


def gat_aggregate(hfeat, a_src, a_dst, graph, H, C, slope=0.2, att_src=None):
    want_grad = torch.is_grad_enabled() and (hfeat.requires_grad or a_src.requires_grad or a_dst.requires_grad)
    return _GATAggregate.apply(hfeat, a_src, a_dst, graph, H, C, slope, att_src, want_grad)


class _GATAttend(torch.autograd.Function):
    """One GATConv attention block as a single autograd node: scores (rgbx_gat_scores_f32) + fused
    edge-softmax/aggregate forward; backward = prep + source pass + segment sum (as _GATAggregate) followed
    by rgbx_gat_scores_bwd_f32, which folds the score gradients into g_hfeat in place and reduces the
    attention-vector gradients without materialising [N, H, C] products. `bias` ([H*C], optional) is added
    in the aggregation kernel's store (GATConv's `+ bias` for concatenated heads). `out_scale` ([H*C], inference
    only — the stored rows are then not the aggregate the backward needs): stored row = aggregate * out_scale + bias,
    an eval-mode BatchNorm after the layer folded into the store."""

    @staticmethod
    def forward(ctx, hfeat, att_src, att_dst, graph, H, C, slope, bias=None, out_scale=None, want_grad=True):
        _lib.require_device(hfeat, att_src, att_dst, bias, out_scale)
        hfeat = hfeat.contiguous()
        b = None if bias is None else bias.detach().reshape(H * C).contiguous()
        sc = None if out_scale is None else out_scale.detach().reshape(H * C).contiguous()
        ctx.inference_only = sc is not None
        att_s = att_src.detach().reshape(H, C).contiguous()
        att_d = att_dst.detach().reshape(H, C).contiguous()
        n_src, N, dev = hfeat.size(0), graph.fwd.N, hfeat.device
        lib = _lib.load()
        ph, ldh = _lib.mat(hfeat, "hfeat")
        want_grad = want_grad and sc is None
        in_kernel = _scores_in_kernel(C)
        if in_kernel:
            # heads spanning <= 8 lanes: both score products are formed inside the aggregation kernel from rows it
            # holds anyway (the gathered row for a_src, the target's own row for a_dst): no scores pass over hfeat.
            # a_dst is stored only for a backward (the per-target records need it)
            a_src = None
            a_dst = torch.empty((N, H), dtype=torch.float32, device=dev) if want_grad else None
        else:
            a_src = torch.empty((n_src, H), dtype=torch.float32, device=dev)
            a_dst = torch.empty_like(a_src)
            _lib.check(lib.rgbx_gat_scores_f32(ph, ldh, _lib.ptr(att_s), _lib.ptr(att_d), _lib.ptr(a_src),
                                               _lib.ptr(a_dst), n_src, H, C, _lib.stream_ptr()), "rgbx_gat_scores_f32")
        out = torch.empty((N, H * C), dtype=torch.float32, device=dev)
        m = torch.empty((N, H), dtype=torch.float32, device=dev)
        rden = torch.empty_like(m)
        po, ldo = _lib.mat(out, "out")
        opos, apos = _gat_train_extras(want_grad, N, H, C, dev)
        split, _scratch = graph.fwd.split_arg((2 * H * C + 3 * H) if want_grad else (H * C + 2 * H), dev)
        with _Timed("gat_fwd"):
            _lib.check(
                lib.rgbx_gat_aggregate_fwd_f32(_lib.ptr(graph.fwd.rowptr), _lib.ptr(graph.fwd.col), ph, ldh,
                                               _lib.ptr(a_src), _lib.ptr(att_s) if in_kernel else None,
                                               _lib.ptr(a_dst), _lib.ptr(att_d) if in_kernel else None, _lib.ptr(sc),
                                               _lib.ptr(b), po, ldo, _lib.ptr(m), _lib.ptr(rden), _lib.ptr(opos),
                                               _lib.ptr(apos), N, H, C, float(slope),
                                               None if split is None else ctypes.byref(split), _lib.stream_ptr()),
                "rgbx_gat_aggregate_fwd_f32")
        ctx.save_for_backward(hfeat, a_src, a_dst, m, rden, out, att_s, att_d, b, opos, apos)
        ctx.graph, ctx.H, ctx.C, ctx.slope = graph, H, C, slope
        ctx.att_shapes = (att_src.shape, att_dst.shape)
        ctx.bias_shape = None if bias is None else bias.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.inference_only:
            raise RuntimeError("gat_attend(out_scale=...) is an inference-only form (eval-mode BatchNorm fold)")
        hfeat, a_src, a_dst, m, rden, out, att_s, att_d, b, opos, apos = ctx.saved_tensors
        H, C = ctx.H, ctx.C
        gout = gout.contiguous()
        if opos is None:
            raise RuntimeError("gat_attend: backward asked of a forward that was run without want_grad")
        g_h, g_as, g_ad = _gat_backward_core(ctx.graph, hfeat, a_src, a_dst, m, rden, out, gout, H, C, ctx.slope, b,
                                             opos=opos, apos=apos, att=(att_s, att_d))
        g_b = gout.sum(0).reshape(ctx.bias_shape) if b is not None and ctx.needs_input_grad[7] else None
        lib = _lib.load()
        n = hfeat.size(0)
        n_scr = ctypes.c_int64(0)
        _lib.check(lib.rgbx_gat_scores_bwd_scratch_floats(n, H, C, ctypes.byref(n_scr)),
                   "rgbx_gat_scores_bwd_scratch_floats")
        scratch = torch.empty(n_scr.value, dtype=torch.float32, device=hfeat.device)
        g_att_s = torch.empty((H, C), dtype=torch.float32, device=hfeat.device)
        g_att_d = torch.empty_like(g_att_s)
        ph, ldh = _lib.mat(hfeat, "hfeat")
        with _Timed("gat_scores_bwd"):  # the source pass folded the score terms into g_h: attention-vector sums only
            _lib.check(
                lib.rgbx_gat_scores_bwd_f32(ph, ldh, _lib.ptr(g_as), _lib.ptr(g_ad), g_ad.size(0), _lib.ptr(att_s),
                                            _lib.ptr(att_d), None, 0, _lib.ptr(g_att_s), _lib.ptr(g_att_d),
                                            _lib.ptr(scratch), n_scr.value, n, H, C, _lib.stream_ptr()),
                "rgbx_gat_scores_bwd_f32")
        return (g_h, g_att_s.reshape(ctx.att_shapes[0]), g_att_d.reshape(ctx.att_shapes[1]), None, None, None, None, g_b,
                None, None)


def gat_attend(h, att_src, att_dst, graph, H, C, slope=0.2, bias=None, out_scale=None):
    """Scores + edge-softmax + aggregation (* out_scale + bias) of one GATConv; dispatches to the partitioned
    graph. `out_scale` is for inference (no_grad) only."""
    if _is_dist(graph):
        out = graph.gat(h, att_src, att_dst, H, C, slope)
        if out_scale is not None:
            out = out * out_scale
        return out if bias is None else out + bias
    if out_scale is not None and torch.is_grad_enabled():
        raise RuntimeError("gat_attend(out_scale=...) is an inference-only form")
    want_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (h, att_src, att_dst, bias))
    return _GATAttend.apply(h, att_src, att_dst, graph, H, C, slope, bias, out_scale, want_grad)


def gat_linear_ok(graph, in_channels, out_channels, x=None, y=None):
    """A single-head GATConv can run aggregate-first (gat_attend_linear): widths the fused kernel's second-aggregate
    form takes, aggregating at the input width not the more expensive order, single-GPU graph."""
    return (not _is_dist(graph) and in_channels in (64, 128, 256) and in_channels <= out_channels <= 128
            and out_channels % 32 == 0 and (x is None or x.is_cuda)
            and (y is None or (y.is_cuda and y.dtype == torch.int64)))


def gat_edge_softmax(csr, a_src, a_dst, slope, train, N):
    """(alpha [E'], alpha_pos or None, m [N], rden [N], a_pos or None) of a single head (rgbx_gat_edge_softmax_f32)."""
    dev = a_src.device
    alpha = torch.empty(max(csr.nnz, 1), dtype=torch.float32, device=dev)
    alpha_pos = torch.empty_like(alpha) if train else None
    m = torch.empty(N, dtype=torch.float32, device=dev)
    rden = torch.empty_like(m)
    a_pos = torch.empty_like(m) if train else None
    split, _scratch = csr.split_arg(1, dev)  # hub rows: a workgroup each (no scratch used)
    with _Timed("gat_edge_softmax"):
        _lib.check(_lib.load().rgbx_gat_edge_softmax_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(a_src),
                                                         _lib.ptr(a_dst), float(slope), _lib.ptr(alpha),
                                                         _lib.ptr(alpha_pos), _lib.ptr(m), _lib.ptr(rden),
                                                         _lib.ptr(a_pos), N,
                                                         None if split is None else ctypes.byref(split),
                                                         _lib.stream_ptr()),
                   "rgbx_gat_edge_softmax_f32")
    return alpha, alpha_pos, m, rden, a_pos


class _GATAttendLinear(torch.autograd.Function):
    """A single-head GATConv with its transform BEHIND the aggregation:
        out_i = (sum_j alpha_ij x_j) W^T + b,   alpha = edge softmax of leaky_relu(a_src[j] + a_dst[i]),
        a_src = x (W^T att_src),  a_dst = x (W^T att_dst)
    — the same function as lin -> scores -> edge softmax -> aggregate -> + bias of GATConv.forward with heads = 1 [PyG]
    (the last layer of reference models/gat.py:21,30), re-associated: sum_j alpha_ij (W x_j) = W sum_j alpha_ij x_j.
    Launches: scores over x (no h = x W^T product), the coefficients as a per-edge vector (rgbx_gat_edge_softmax_f32;
    one head: 4 bytes per edge), then rgbx_fused_layer_f32 with w = alpha — the transform on the MFMA units under the
    gather and, with labels given, the loss inside the kernel (no logits). A forward that prepares a backward also
    stores the aggregate z (dW = dy^T z) and its positive-score part (w_pos): exactly the (out, out_pos) pair the
    per-node GAT backward needs with hfeat := x and gout := dy W, so the backward is that of _GATAttend on those
    operands plus the chain through v = att W. `y` / `mask`: returns (loss, stats) as _PropagateLinearCE does."""

    @staticmethod
    def forward(ctx, x, weight, att_src, att_dst, bias, graph, slope, y, mask, want_grad):
        _lib.require_device(x, weight, att_src, att_dst, bias)
        x = x.contiguous()
        W = weight.detach()
        C, K = W.shape
        att = torch.stack([att_src.detach().reshape(C), att_dst.detach().reshape(C)])  # [2, C]
        v = (att @ W).contiguous()  # [2, K]: the scores are products of x with these vectors
        n_src, N, dev = x.size(0), graph.fwd.N, x.device
        a_src = torch.empty(n_src, dtype=torch.float32, device=dev)
        a_dst = torch.empty_like(a_src)
        px, ldx = _lib.mat(x, "x")
        _lib.check(_lib.load().rgbx_gat_scores_f32(px, ldx, _lib.ptr(v[0]), _lib.ptr(v[1]), _lib.ptr(a_src),
                                                   _lib.ptr(a_dst), n_src, 1, K, _lib.stream_ptr()), "rgbx_gat_scores_f32")
        alpha, alpha_pos, m, rden, a_pos = gat_edge_softmax(graph.fwd, a_src, a_dst, slope, want_grad, N)
        ce = None
        if y is not None:
            ce = (y, mask, mask_scale(y, mask, C) if want_grad else None)
        out, zz, stats = fused_layer(x, weight_t(weight), csr=graph.fwd, w=alpha,
                                     bias=None if bias is None else bias.detach(), ce=ce, kind="gat_linear_fwd",
                                     w_pos=alpha_pos)
        ctx.graph, ctx.slope, ctx.has_bias, ctx.ce = graph, slope, bias is not None, y is not None
        ctx.shapes = (att_src.shape, att_dst.shape)
        if want_grad:
            z, z_pos = zz
            ctx.save_for_backward(x, a_src, a_dst, m, rden, z, z_pos, a_pos, v, weight, att, out if ctx.ce else None)
        if y is None:
            return out
        ctx.mark_non_differentiable(stats)
        return (stats[0] / stats[1]).float(), stats

    @staticmethod
    def backward(ctx, g, _g_stats=None):
        if not ctx.saved_tensors:
            raise RuntimeError("gat_attend_linear: backward asked of a forward that was run without want_grad")
        x, a_src, a_dst, m, rden, z, z_pos, a_pos, v, weight, att, dlogits = ctx.saved_tensors
        W = weight.detach()
        C, K = W.shape
        if ctx.ce:  # the kernel stored the loss gradient w.r.t. the logits; the incoming scalar rides on W and dW
            dy, scalar = dlogits, g.reshape(()).float()
            Wg = W * scalar
        else:
            dy, scalar, Wg = g.contiguous(), None, W
        g_z = dy @ Wg  # [N, K]: gradient w.r.t. the aggregate
        gw, gcol = gemm_tn(dy, z, colsum=True)
        g_x, g_as, g_ad = _gat_backward_core(ctx.graph, x, a_src, a_dst, m, rden, z, g_z, 1, K, ctx.slope, opos=z_pos,
                                             apos=a_pos, att=(v[0:1], v[1:2]))
        lib = _lib.load()
        n = x.size(0)
        n_scr = ctypes.c_int64(0)
        _lib.check(lib.rgbx_gat_scores_bwd_scratch_floats(n, 1, K, ctypes.byref(n_scr)), "rgbx_gat_scores_bwd_scratch_floats")
        scratch = torch.empty(n_scr.value, dtype=torch.float32, device=x.device)
        g_v = torch.empty((2, K), dtype=torch.float32, device=x.device)
        px, ldx = _lib.mat(x, "x")
        with _Timed("gat_scores_bwd"):  # the source pass folded the score terms into g_x: the vectors' sums only
            _lib.check(
                lib.rgbx_gat_scores_bwd_f32(px, ldx, _lib.ptr(g_as), _lib.ptr(g_ad), g_ad.size(0), _lib.ptr(v[0]),
                                            _lib.ptr(v[1]), None, 0, _lib.ptr(g_v[0]), _lib.ptr(g_v[1]),
                                            _lib.ptr(scratch), n_scr.value, n, 1, K, _lib.stream_ptr()),
                "rgbx_gat_scores_bwd_f32")
        # v = att W: g_att = g_v W^T, and W collects att^T g_v next to dy^T z
        g_att = g_v @ W.t()
        if scalar is not None:
            gw, gcol = gw * scalar, gcol * scalar
        gw = torch.addmm(gw, att.t(), g_v)
        need = ctx.needs_input_grad
        return (g_x if need[0] else None, gw if need[1] else None,
                g_att[0].reshape(ctx.shapes[0]) if need[2] else None, g_att[1].reshape(ctx.shapes[1]) if need[3] else None,
                gcol if ctx.has_bias and need[4] else None, None, None, None, None, None)


def gat_attend_linear(x, weight, att_src, att_dst, graph, slope=0.2, bias=None, ce=None):
    """Single-head GATConv, aggregate-first (the caller checked gat_linear_ok). `ce` = (y, mask): returns
    (loss, stats) with the loss taken inside the kernel; else the logits."""
    want_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (x, weight, att_src, att_dst, bias))
    y, mask = ce if ce is not None else (None, None)
    return _GATAttendLinear.apply(x, weight, att_src, att_dst, bias, graph, slope, y, mask, want_grad)


def _scores_in_kernel(C):
    """Forming <h_j, att_src> from the gathered row costs log2(lanes per head) cross-lane adds per
    neighbour and saves the a_src[j] cache-line request. Measured at |V|=2M, |E|=60M: 4 lanes per head
    (H=8, C=16) 6.20 -> 5.51 ms; 32 lanes per head (H=1, C=128) 5.71 -> 6.02 ms. Use it up to 8 lanes."""
    vec = 4 if C % 4 == 0 else (2 if C % 2 == 0 else 1)
    lanes = 1
    while lanes * vec < C:
        lanes *= 2
    return lanes <= 8


def gather_rows(src, idx, out=None):
    """out[r] = src[idx[r]] (idx int32, device)."""
    _lib.require_device(src, idx)
    ps, lds = _lib.mat(src, "src")
    n, d = idx.numel(), src.size(1)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=src.device)
    po, ldo = _lib.mat(out, "dst")
    _lib.check(_lib.load().rgbx_gather_rows_f32(ps, lds, _lib.ptr(idx), n, d, po, ldo, _lib.stream_ptr()),
               "rgbx_gather_rows_f32")
    return out


def scatter_add_rows(src, idx, dst):
    """dst[idx[r]] += src[r]; idx entries unique."""
    _lib.require_device(src, idx, dst)
    ps, lds = _lib.mat(src, "src")
    pd, ldd = _lib.mat(dst, "dst")
    _lib.check(
        _lib.load().rgbx_scatter_add_rows_f32(ps, lds, _lib.ptr(idx), idx.numel(), src.size(1), pd, ldd,
                                              _lib.stream_ptr()), "rgbx_scatter_add_rows_f32")
    return dst


# ---- dense layers: forward through hipBLASLt, weight gradient through the split-K MFMA kernel --------

def gemm_tn(a, b, alpha=1.0, colsum=False, out=None, sums_out=None):
    """a^T b for a [K,M], b [K,N] (fp32, device): rgbx_gemm_tn_f32. With `colsum` also the column sums of `a`
    ([M], from the same pass): returns (a^T b, a.sum(0)). `out` ([M, N] contiguous) / `sums_out` ([M]): write there
    (e.g. views of one flat gradient buffer)."""
    _lib.require_device(a, b)
    a = a if a.stride(-1) == 1 and _rows_padded_readable(a) else a.contiguous()
    b = b if b.stride(-1) == 1 and _rows_padded_readable(b) else b.contiguous()
    # rows off the 16-byte grid (width % 4 != 0 at its natural stride: dY [N, 7], a dropout copy of [N, 1433] features) take
    # the kernel's 4-byte path: 16.5 ms against 5.0 for 2 M x 64 x 1433, 0.83 against 0.37 for 2 M x 7 x 64. One padded copy
    # (align_rows: a read and a write of the operand) buys the 16-byte path — 3 ms for the 11.5 GB operand.
    if a.size(0) >= 4096:
        a, b = (_aligned_rows(t) for t in (a, b))
    pa, lda = _lib.mat(a, "a")
    pb, ldb = _lib.mat(b, "b")
    K, M, N = a.size(0), a.size(1), b.size(1)
    lib = _lib.load()
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.rgbx_gemm_tn_workspace_bytes(K, M, N, ctypes.byref(nbytes)), "rgbx_gemm_tn_workspace_bytes")
    ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=a.device)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    elif tuple(out.shape) != (M, N) or not out.is_contiguous() or out.dtype != torch.float32:
        raise RuntimeError(f"gemm_tn: out must be a contiguous float32 [{M}, {N}] tensor")
    sums = None
    if colsum:
        sums = sums_out if sums_out is not None else torch.empty(M, dtype=torch.float32, device=a.device)
        if tuple(sums.shape) != (M,) or not sums.is_contiguous() or sums.dtype != torch.float32:
            raise RuntimeError(f"gemm_tn: sums_out must be a contiguous float32 [{M}] tensor")
    with _Timed("gemm_tn", f"{M}x{N}" if _EVENT_SINK is not None else None):
        _lib.check(lib.rgbx_gemm_tn_f32(pa, lda, pb, ldb, _lib.ptr(out), N, _lib.ptr(sums), K, M, N, float(alpha),
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "rgbx_gemm_tn_f32")
    return (out, sums) if colsum else out


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b). dW = dy^T x runs on rgbx_gemm_tn_f32 when the tensors are on the GPU."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if _EVENT_SINK is None or not x.is_cuda:
            return torch.nn.functional.linear(x, weight, bias)
        with _Timed("linear_fwd", f"{weight.size(1)}->{weight.size(0)}"):  # hipBLASLt; timed only for bench.py's tables
            return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and gy.is_cuda and want_b:
            gw, gb = gemm_tn(gy, x, colsum=True)  # dy is read once for dW and db
        else:
            if ctx.needs_input_grad[1]:
                gw = gemm_tn(gy, x) if gy.is_cuda else gy.t() @ x
            gb = gy.sum(0) if want_b else None
        return gx, gw, gb


def linear(x, weight, bias=None):
    sp = getattr(x, "_rgbx_sparse", None)
    if sp is not None and not x.requires_grad:  # static features with few non-zeros (prepare_features)
        return _SparseRowsLinear.apply(sp, weight, bias)
    return _Linear.apply(x, weight, bias)


SHORT_ROWS_TABLE_BYTES = 2 << 20  # half of one XCD's 4 MB L2: the gathered table must stay cache-resident beside the streams
SHORT_ROWS_MEAN_SLOTS = 64


def short_rows_ok(csr, table):
    """rgbx_spmm_csr_short_rows_f32 (a lane group per target row) is the better form: rows of tens of slots over a table that
    stays in L2 — the product over a bag-of-words matrix's non-zeros. Measured at N = 2 M, 36 M non-zeros (profiles/
    r05_short_rows.txt)."""
    d = table.size(1)
    # d <= 64: beyond, both forms read nnz * d * 4 bytes from L2 at its rate (d = 128: 1.13 ms row per wave, 1.20 ms here)
    return (table.is_cuda and d % 4 == 0 and 4 <= d <= 64 and table.numel() * 4 <= SHORT_ROWS_TABLE_BYTES
            and csr.nnz <= SHORT_ROWS_MEAN_SLOTS * max(csr.N, 1))


def spmm_short_rows_raw(csr, w, x, bias=None, kind="spmm"):
    """out[i,:] = sum_p w[p] x[col[p],:] (+ bias) on rgbx_spmm_csr_short_rows_f32 (no autograd)."""
    _lib.require_device(x, w, bias)
    px, ldx = _lib.mat(x, "x")
    N, d = csr.N, x.size(1)
    out = torch.empty((N, d), dtype=torch.float32, device=x.device)
    with _Timed(kind, f"shortrows+d{d}" if _EVENT_SINK is not None else None):
        _lib.check(_lib.load().rgbx_spmm_csr_short_rows_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(w), px, ldx,
                                                            _lib.ptr(bias), _lib.ptr(out), d, N, d, _lib.stream_ptr()),
                   "rgbx_spmm_csr_short_rows_f32")
    return out


class SparseFeatures:
    """What ops.dropout returns for features that ride on their non-zeros: not a tensor — only ops.linear (nn.Linear of this
    package) takes it; anything else fails loudly instead of reading undropped dense values."""
    requires_grad = False

    def __init__(self, sp, like):
        self._rgbx_sparse, self.shape, self.device, self.dtype, self.is_cuda = sp, like.shape, like.device, like.dtype, True

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return len(self.shape)


def dropout(x, p, training):
    """F.dropout for a model's INPUT features (reference models/dagnn.py:72: dropout before the first Linear). Dense
    features: torch's. Features carried by their non-zeros (prepare_features) in a training forward: dropout leaves a zero a
    zero, so the Bernoulli mask is drawn for the non-zeros alone — the same distribution of outputs, from nnz draws instead
    of a fresh dense [N, F] matrix per step (11.5 GB written and read again at N = 2 M, F = 1433) — and the product stays on
    the non-zeros. (The mask comes from torch's device generator: reproducible under manual_seed; no implementation on
    another device reproduces the reference's CUDA Philox stream anyway.)"""
    sp = getattr(x, "_rgbx_sparse", None)
    if sp is None or not training or p <= 0.0 or x.requires_grad:
        return torch.nn.functional.dropout(x, p=p, training=training)
    if p >= 1.0:
        return SparseFeatures(sp.with_values(torch.zeros_like(sp.val)), x)
    keep = torch.rand(sp.val.numel(), device=sp.val.device) >= p
    return SparseFeatures(sp.with_values(sp.val * keep * (1.0 / (1.0 - p))), x)


class SparseRows:
    """The non-zeros of a STATIC feature matrix [n, f] as a CSR (rows -> (column, value)) and its transpose (columns ->
    (row, value)), built once per matrix with torch index ops (plumbing: one pass over the matrix and one stable sort of the
    non-zeros). The reference's datasets are bag-of-words (Cora: 18 non-zeros of F = 1433 per row) that it multiplies as
    dense matrices; x W^T and dW = dY^T x over the non-zeros alone are two launches of the row-gather kernel
    (rgbx_spmm_csr_f32: the weights are the feature values, the gathered table W^T [f, out] resp. dY [n, out]) and read
    nnz * out * 4 bytes instead of n * f * 4 — the same sums up to float32 rounding order (zeros contribute exactly 0)."""

    def __init__(self, x):
        from .graph import CSR, make_row_split
        _lib.require_device(x)
        n, f = x.shape
        # row-major order, in row blocks of at most 2^28 entries: torch's nonzero() indexes with 32 bits (2 M x 1433 entries
        # in one call ended in an allocation of 2^56 bytes, round 5)
        step = max(1, (1 << 28) // max(f, 1))
        rows, cols, vals = [], [], []
        for r0 in range(0, n, step):
            blk = x[r0:r0 + step]
            nz = blk.nonzero(as_tuple=False)
            rows.append(nz[:, 0] + r0)
            cols.append(nz[:, 1].contiguous())
            vals.append(blk[nz[:, 0], nz[:, 1]])
        rows, cols, vals = (torch.cat(t).contiguous() if t else x.new_zeros(0, dtype=dt)
                            for t, dt in ((rows, torch.int64), (cols, torch.int64), (vals, x.dtype)))
        self.n, self.f, self.nnz = n, f, int(rows.numel())
        if self.nnz >= 2 ** 31 - 1:
            raise RuntimeError("SparseRows: more than 2^31 non-zeros")
        i32 = lambda t: t.to(torch.int32).contiguous()

        def rowptr_of(idx, m):
            ptr = torch.zeros(m + 1, dtype=torch.int64, device=x.device)
            ptr[1:] = torch.cumsum(torch.bincount(idx, minlength=m), 0)
            return i32(ptr)

        pad = lambda t: t if t.numel() else t.new_zeros(1)  # (the kernels take a non-null pointer)
        ptr = rowptr_of(rows, n)
        self.fwd = CSR(ptr, pad(i32(cols)), None, n, self.nnz, make_row_split(ptr))
        self.val = pad(vals)
        order = torch.argsort(cols, stable=True)
        self.order = order  # slot of the transposed CSR -> slot of the forward CSR
        ptr_t = rowptr_of(cols, f)
        self.bwd = CSR(ptr_t, pad(i32(rows[order])), None, f, self.nnz, make_row_split(ptr_t))
        self.val_t = pad(vals[order].contiguous())

    def with_values(self, vals):
        """The same structure (shared, not copied) under other values per non-zero, given in the forward CSR's slot order."""
        import copy
        other = copy.copy(self)
        other.val = vals.contiguous()
        other.val_t = vals[self.order].contiguous() if self.nnz else vals
        return other


class _SparseRowsLinear(torch.autograd.Function):
    """y = x W^T (+ b) and dW = dY^T x, db = column sums of dY, for x given by its non-zeros (SparseRows). x takes no gradient."""

    @staticmethod
    def forward(ctx, sp, weight, bias):
        ctx.sp, ctx.has_bias = sp, bias is not None
        wt = weight.detach().t().contiguous()  # [f, out]: the gathered table (L2-resident: 367 KB for 1433 x 64)
        b = None if bias is None else bias.detach().contiguous()
        if short_rows_ok(sp.fwd, wt):
            return spmm_short_rows_raw(sp.fwd, sp.val, wt, bias=b, kind="features_fwd")
        return spmm_raw(sp.fwd, sp.val, None, wt, kind="features_fwd", bias=b)

    @staticmethod
    def backward(ctx, gy):
        sp = ctx.sp
        gy = gy.contiguous()
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = spmm_raw(sp.bwd, sp.val_t, None, gy, kind="features_bwd").t()  # [f, out] -> dW [out, f]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return None, gw, gb


def prepare_features(x, max_density=0.1):
    """The layout experiment() gives the static input features on the GPU: rows on 16-byte boundaries (align_rows) and,
    when at most `max_density` of the entries are non-zero (bag-of-words features: Cora 1.3 %), their non-zeros as a
    SparseRows riding on the tensor — ops.linear then multiplies over the non-zeros alone. Same values, same shape."""
    x = align_rows(x)
    if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and x.numel():
        step = max(1, (1 << 28) // x.size(1))  # (row blocks: see SparseRows)
        nnz = sum(int(torch.count_nonzero(x[r0:r0 + step])) for r0 in range(0, x.size(0), step))
        if nnz <= max_density * x.numel() and nnz < 2 ** 31 - 1:
            x._rgbx_sparse = SparseRows(x)  # (this Python object only: a slice or a copy is an ordinary dense tensor)
    return x


# ---- masked NLL + accuracy ------------------------------------------------------------------------

def _nll_stats(logp, y, mask, want_acc):
    _lib.require_device(logp, y, mask)
    if y.dtype != torch.int64:
        raise RuntimeError(f"labels must be int64, got {y.dtype}")
    if mask is not None and mask.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError(f"mask must be bool, got {mask.dtype}")
    pl, ld = _lib.mat(logp, "logp")
    lib = _lib.load()
    stats = torch.empty(3, dtype=torch.float64, device=logp.device)
    n_scr = ctypes.c_int64(0)
    _lib.check(lib.rgbx_masked_nll_scratch_doubles(logp.size(0), int(want_acc), ctypes.byref(n_scr)),
               "rgbx_masked_nll_scratch_doubles")
    scratch = torch.empty(n_scr.value, dtype=torch.float64, device=logp.device)
    y = y.contiguous()
    mask = None if mask is None else mask.contiguous()
    _lib.check(lib.rgbx_masked_nll_fwd_f32(pl, ld, _lib.ptr(y), _lib.ptr(mask), logp.size(0), logp.size(1),
                                           _lib.ptr(stats), _lib.ptr(scratch), n_scr.value, int(want_acc),
                                           _lib.stream_ptr()), "rgbx_masked_nll_fwd_f32")
    return stats, y, mask


class _MaskedNLL(torch.autograd.Function):
    """nn.NLLLoss()(logp[mask], y[mask]) without materialising the selection (itexperiments.py:400,429).
    reduction 'mean' divides by the number of selected rows, 'sum' does not."""

    @staticmethod
    def forward(ctx, logp, y, mask, reduction):
        logp = logp if logp.stride(-1) == 1 else logp.contiguous()
        stats, y, mask = _nll_stats(logp, y, mask, False)
        ctx.y, ctx.mask, ctx.shape, ctx.reduction = y, mask, logp.shape, reduction
        ctx.count = stats[1]
        return (stats[0] / stats[1] if reduction == "mean" else stats[0]).float()

    @staticmethod
    def backward(ctx, g):
        scale = (g.double() / ctx.count if ctx.reduction == "mean" else g.double()).float().reshape(1).contiguous()
        N, C = ctx.shape
        grad = torch.empty((N, C), dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().rgbx_masked_nll_bwd_f32(_lib.ptr(ctx.y), _lib.ptr(ctx.mask), N, C, _lib.ptr(scale),
                                                       _lib.ptr(grad), C, _lib.stream_ptr()),
                   "rgbx_masked_nll_bwd_f32")
        return grad, None, None, None


def masked_nll_loss(logp, y, mask=None, reduction="mean"):
    return _MaskedNLL.apply(logp, y, mask, reduction)


def _ce_stats(logits, y, mask):
    _lib.require_device(logits, y, mask)
    if y.dtype != torch.int64:
        raise RuntimeError(f"labels must be int64, got {y.dtype}")
    if mask is not None and mask.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError(f"mask must be bool, got {mask.dtype}")
    pz, ld = _lib.mat(logits, "logits")
    lib = _lib.load()
    stats = torch.empty(3, dtype=torch.float64, device=logits.device)
    n_scr = ctypes.c_int64(0)
    _lib.check(lib.rgbx_masked_nll_scratch_doubles(logits.size(0), 1, ctypes.byref(n_scr)),
               "rgbx_masked_nll_scratch_doubles")
    scratch = torch.empty(n_scr.value, dtype=torch.float64, device=logits.device)
    y = y.contiguous()
    mask = None if mask is None else mask.contiguous()
    _lib.check(lib.rgbx_masked_ce_fwd_f32(pz, ld, _lib.ptr(y), _lib.ptr(mask), logits.size(0), logits.size(1),
                                          _lib.ptr(stats), _lib.ptr(scratch), n_scr.value, _lib.stream_ptr()),
               "rgbx_masked_ce_fwd_f32")
    return stats, y, mask


class _MaskedCE(torch.autograd.Function):
    """NLLLoss(log_softmax(z)[mask], y[mask]) taken from the logits z: the loss, its gradient
    scale * (softmax - onehot) and (stats[2]) the arg-max hits, without writing log-softmax or a one-hot
    gradient. Returns (loss, stats [3] float64: nll sum, selected rows, correct)."""

    @staticmethod
    def forward(ctx, logits, y, mask, reduction):
        logits = logits if logits.stride(-1) == 1 else logits.contiguous()
        stats, y, mask = _ce_stats(logits, y, mask)
        ctx.save_for_backward(logits)
        ctx.y, ctx.mask, ctx.reduction = y, mask, reduction
        ctx.count = stats[1]
        ctx.mark_non_differentiable(stats)
        return (stats[0] / stats[1] if reduction == "mean" else stats[0]).float(), stats

    @staticmethod
    def backward(ctx, g, _g_stats):
        (logits,) = ctx.saved_tensors
        scale = (g.double() / ctx.count if ctx.reduction == "mean" else g.double()).float().reshape(1).contiguous()
        N, C = logits.shape
        grad = torch.empty((N, C), dtype=torch.float32, device=g.device)
        pz, ld = _lib.mat(logits, "logits")
        _lib.check(_lib.load().rgbx_masked_ce_bwd_f32(pz, ld, _lib.ptr(ctx.y), _lib.ptr(ctx.mask), N, C,
                                                      _lib.ptr(scale), _lib.ptr(grad), C, _lib.stream_ptr()),
                   "rgbx_masked_ce_bwd_f32")
        return grad, None, None, None


def masked_ce_loss(logits, y, mask=None, reduction="mean", with_stats=False):
    """Cross-entropy of the raw logits on the masked rows = masked_nll_loss(log_softmax(logits), ...), in one
    pass each way. `with_stats`: also the [nll sum, count, correct] tensor of the same pass."""
    loss, stats = _MaskedCE.apply(logits, y, mask, reduction)
    return (loss, stats) if with_stats else loss


def masked_ce_accuracy(logits, y, mask=None):
    """masked_nll_accuracy(log_softmax(logits), ...) from the logits: float64 device tensor [3]."""
    logits = logits.detach()
    logits = logits if logits.stride(-1) == 1 else logits.contiguous()
    return _ce_stats(logits, y, mask)[0]


def masked_ce_accuracy_blocked(blk, y, mask=None, bias=None):
    """masked_ce_accuracy of logits held BLOCKED ([B, n, C / B]: the column slices a node-partitioned run's exchange
    delivers) plus `bias` per column, read in place — only the selected rows are touched (rgbx_masked_ce_fwd_blocked_f32)."""
    _lib.require_device(blk, y, mask, bias)
    if y.dtype != torch.int64:
        raise RuntimeError(f"labels must be int64, got {y.dtype}")
    if mask is not None and mask.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError(f"mask must be bool, got {mask.dtype}")
    ptr, bc, bs = _blocked(blk, "logits")
    n, C = blk.size(1), blk.size(0) * blk.size(2)
    lib = _lib.load()
    stats = torch.empty(3, dtype=torch.float64, device=blk.device)
    n_scr = ctypes.c_int64(0)
    _lib.check(lib.rgbx_masked_nll_scratch_doubles(n, 1, ctypes.byref(n_scr)), "rgbx_masked_nll_scratch_doubles")
    scratch = torch.empty(n_scr.value, dtype=torch.float64, device=blk.device)
    y = y.contiguous()
    mask = None if mask is None else mask.contiguous()
    b = None if bias is None else bias.detach().contiguous()
    _lib.check(lib.rgbx_masked_ce_fwd_blocked_f32(ptr, bc, bs, _lib.ptr(b), _lib.ptr(y), _lib.ptr(mask), n, C,
                                                  _lib.ptr(stats), _lib.ptr(scratch), n_scr.value, _lib.stream_ptr()),
               "rgbx_masked_ce_fwd_blocked_f32")
    return stats


def masked_nll_accuracy(logp, y, mask=None):
    """(sum of -logp[i,y_i], selected-row count, correct arg-max count) as a float64 device tensor [3]."""
    logp = logp.detach()
    logp = logp if logp.stride(-1) == 1 else logp.contiguous()
    return _nll_stats(logp, y, mask, True)[0]
