"""Shared skeleton of the reference's conv-stack models: (conv -> BatchNorm1d) x (L-1), conv,
log_softmax. No activation and no dropout are applied (the reference stores ``dropout_rate`` but
never uses it: models/gcn.py:15,25-31)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn import BatchNorm1d


class ModelOutput(dict):
    """The reference's forward contract, {'out': log-probs, 'emb': logits} (models/gcn.py:31), plus 'x' as an
    alias of 'out' (README.md:51 names the key 'x' while itexperiments.py:428 reads 'out').
    'out' / 'x' are produced on first access: a caller that reads res['out'] (the reference's loop, test())
    gets F.log_softmax(logits, dim=1) exactly as before; the build's own loops take the loss, its gradient and
    the accuracy straight from 'emb' (ops.masked_ce_*) and never pay for writing [N, C] log-probabilities."""

    _LAZY = ("out", "x")

    def __init__(self, logits):
        super().__init__(emb=logits)

    def _log_probs(self):
        out = F.log_softmax(dict.__getitem__(self, "emb"), dim=1)
        dict.__setitem__(self, "out", out)
        dict.__setitem__(self, "x", out)
        return out

    def __missing__(self, key):
        if key in self._LAZY:
            return self._log_probs()
        raise KeyError(key)

    def get(self, key, default=None):
        if key in self._LAZY and not dict.__contains__(self, key):
            return self._log_probs()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key in self._LAZY or dict.__contains__(self, key)

    def keys(self):
        return ["out", "emb", "x"]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return 3

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def copy(self):
        return dict(self.items())

    def pop(self, key, *default):
        if key in self._LAZY and not dict.__contains__(self, key):
            self._log_probs()
        return dict.pop(self, key, *default)

    def __eq__(self, other):
        return dict(self.items()) == other

    __hash__ = None

    def __repr__(self):
        return repr(dict(self.items()))


def model_output(logits):
    return ModelOutput(logits)


class ConvStack(nn.Module):
    """`widths` = [in, hid, ..., hid, out]; `make_conv(i, fan_in, fan_out)` builds layer i;
    `bn_width` = feature width after every non-final conv."""

    def __init__(self, num_layers, dropout_rate, widths, make_conv, bn_width):
        super().__init__()
        if num_layers < 2:
            raise ValueError("num_layers must be at least 2")
        self.num_layers = num_layers
        self.dropout_rate = dropout_rate
        self.convs = nn.ModuleList(make_conv(i, widths[i], widths[i + 1]) for i in range(num_layers))
        self.bns = nn.ModuleList(BatchNorm1d(bn_width) for _ in range(num_layers - 1))

    def forward(self, x, edge_index):
        return model_output(self._run(x, edge_index))

    def masked_ce(self, x, edge_index, y, mask):
        """(loss, stats) = NLLLoss(log_softmax(logits)[mask], y[mask]) and [nll sum, selected rows, correct] of this
        model's logits — what the training loop and the metrics need of a forward (itexperiments.py:429,434,624-626) —
        with the last conv taking the loss into its kernel where it can: the logits are then never written
        (ops.propagate_linear_ce). Otherwise the same numbers from the materialised logits. Under no_grad in eval mode
        the loss itself may be None (nobody reads it there): it is stats[0] / stats[1]."""
        return self._run(x, edge_index, ce=(y, mask))

    def masked_ce_pair(self, x, edge_index, y, mask_a, mask_b):
        """[2, 3] statistics ([nll sum, selected rows, correct] under mask_a and under mask_b) of ONE eval forward: the val
        and the test metrics of an epoch, for which the reference runs two identical eval forwards
        (itexperiments.py:464-473). Where the last conv takes the loss into its kernel both sets come out of that one
        launch (rgbx_ce_epilogue_t.mask_groups = 2) and the logits are never written; otherwise from the logits."""
        return self._run(x, edge_index, ce=(y, (mask_a, mask_b)))[1]

    def _run(self, x, edge_index, ce=None):
        """x = bns[i](convs[i](x)) ... convs[-1](x) (models/gcn.py:25-31). A BatchNorm is never a pass of its own
        where a neighbouring conv can absorb it: under no_grad its eval-mode affine map goes into the PRECEDING conv's
        weights; in a training forward it is handed to the FOLLOWING conv (forward_after_bn), whose fused kernel
        applies it to the aggregate of the raw rows."""
        from .. import ops
        last = self.num_layers - 1
        pending = None  # a BatchNorm whose output has not been formed yet
        if self.training and torch.is_grad_enabled():
            self._train_forwards = getattr(self, "_train_forwards", 0) + 1  # statistics and weights are about to move
        folded = self._eval_operands() if (not torch.is_grad_enabled() and not self.training and x.is_cuda) else None
        agg0 = self._input_aggregate(x, edge_index) if getattr(self, "cache_input_aggregate", False) else None
        if folded is not None and agg0 is None and getattr(self, "collapse_eval", True):
            out = self._run_collapsed(x, edge_index, ce)
            if out is not None:
                return out
        for i, conv in enumerate(self.convs):
            bn = self.bns[i] if i < last else None
            if i == 0 and agg0 is not None and i < last:
                # OPT-IN (cache_input_aggregate): P x of the static input features is the same in every forward of
                # every epoch; with it kept, the first layer is a dense launch over the kept aggregate
                if folded is not None and folded[0] is not None:
                    x = conv.forward_from_aggregate(agg0, x, folded=folded[0])
                    continue
                if torch.is_grad_enabled() or bn is None or bn.training:
                    want = (bn is not None and bn.training and torch.is_grad_enabled()
                            and hasattr(bn, "begin_training_step"))
                    x, pending = conv.forward_from_aggregate(agg0, x, want_colsums=want), bn
                    continue
            if folded is not None and pending is None and folded[i] is not None:
                # eval forward from prepared operands: one fused launch per layer, no weight arithmetic
                res = conv.forward_folded(x, edge_index, folded[i], ce=ce if i == last else None)
                if res is not None:
                    x = res
                    continue
            # a training-mode BatchNorm follows a conv that can hand it its column sums (taken from the fused kernel's
            # MFMA tiles): no statistics pass over the conv's output
            extra = {}
            if (bn is not None and bn.training and torch.is_grad_enabled() and getattr(conv, "emits_colsums", False)
                    and hasattr(bn, "begin_training_step")):
                extra["want_colsums"] = True
            if (i == last and ce is not None and getattr(conv, "accepts_ce", False)
                    and (not isinstance(ce[1], (tuple, list)) or getattr(conv, "accepts_ce_pair", False))):
                extra["ce"] = ce  # the last conv returns (loss, stats) instead of the logits (a pair of masks: (None, [2, 3]))
            if pending is not None:
                sums = getattr(x, ops.COLSUMS, None)
                after = getattr(conv, "forward_after_bn", None)
                if after is not None:  # = conv(pending(x), edge_index)
                    x, pending = after(x, edge_index, pending, colsums=sums, **extra), bn
                    continue
                x, pending = (pending(x, colsums=sums) if sums is not None else pending(x)), None
            affine = None
            if bn is not None and getattr(conv, "folds_post_affine", False) and not torch.is_grad_enabled():
                fold = getattr(bn, "eval_affine", None)
                affine = fold() if fold is not None else None
            if affine is not None:  # eval forward: BatchNorm's affine map folded into the conv's weights
                x = conv(x, edge_index, post_affine=affine)
            else:
                x, pending = conv(x, edge_index, **extra), bn
        if ce is not None and not isinstance(x, tuple):  # a last conv without the loss epilogue (GATConv)
            x = ops.ce_from_logits(x, ce[0], ce[1])
        return x


def _layer_maps(conv):
    """(W_l [out, in], W_r [out, in] or None, b [out], aggregation kind, self-loop rewrite) of a conv layer that is
    `A(x) W_l^T + x W_r^T + b` with a row-wise aggregation A, or None: GCNConv (A = A_hat, no root term), my_SAGEConv (mean
    over N(i) and i itself — its b_l sits inside the mean, whose rows sum to 1: models/graphsage.py:49-62), SAGEConv (mean over
    N(i), lin_l's bias outside it [PyG])."""
    from ..graph import LOOPS_ADD_REMAINING, LOOPS_KEEP, LOOPS_REMOVE_ADD
    name = type(conv).__name__
    if name == "GCNConv":
        return conv.lin.weight, None, conv.bias, "gcn", LOOPS_ADD_REMAINING
    if name == "MySAGEConv" and conv.add_self_loops:
        return conv.lin_l.weight, conv.lin_r.weight, conv.lin_l.bias + conv.lin_r.bias, "mean", LOOPS_REMOVE_ADD
    if name == "SAGEConv":
        return conv.lin_l.weight, conv.lin_r.weight, conv.lin_l.bias, "mean", LOOPS_KEEP
    return None


def _collapsed_operands(self):
    """A conv stack in eval mode is a POLYNOMIAL in its aggregation operator: the reference's stacks apply no activation between
    their layers (models/gcn.py:25-31, graphsage.py:26-32, graphsage2.py:27-33: conv -> BatchNorm -> conv; dropout_rate is stored
    and never used), an eval-mode BatchNorm is a per-column affine map x * s + t, and A (A_hat, or the mean) acts on rows while
    the weights act on columns — the two commute. With layer l = A(h) W_l^T + h W_r^T + b_l:
        logits = sum_k A^k X Q_k + sum_k A^k 1 d_k^T,    Q_k [F, C], d_k [C]: sums over the weight paths that aggregate k times,
    evaluated by Horner's rule: U = X Q_L; U = A(U) + X Q_k + d_k for k = L-1 .. 0. Every aggregation then runs at the CLASS
    width (C = 7 -> rows of 8 floats) instead of the hidden width (64), and the input meets ONE product [N, F] x [F, (L+1) C']
    (GCN: W_r = 0, only Q_L is there). Recurrence per layer (S = diag(s), identity behind the last layer):
        Q'_k = Q_k W_r^T S + Q_{k-1} W_l^T S,   d'_k = d_k W_r^T S + d_{k-1} W_l^T S,   d'_0 += b * s + t.
    Returns (weight [(blocks * C'), F] for ops.linear, which k have a block, [d_k padded or None], C, C', kind, loops_mode) or
    None where this does not apply or does not pay: other layer types, mixed aggregations, a BatchNorm without the affine eval
    form, C' not below every hidden width. Cached per model state like _eval_operands."""
    from .. import ops
    tensors = list(self.parameters()) + list(self.buffers())
    key = (getattr(self, "_train_forwards", 0), ops.weights_epoch()) + tuple((t._version, t.data_ptr()) for t in tensors)
    cached = getattr(self, "_collapsed", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    out = None
    convs, last = list(self.convs), self.num_layers - 1
    maps = [_layer_maps(c) for c in convs]
    C = getattr(convs[-1], "out_channels", 0)
    Cp = (C + 3) // 4 * 4
    if (all(m is not None for m in maps) and len({(m[3], m[4]) for m in maps}) == 1 and 0 < Cp <= 256
            and all(c.out_channels > Cp for c in convs[:-1])):
        affines = [getattr(self.bns[i], "eval_affine", lambda: None)() for i in range(last)]
        if all(a is not None for a in affines):
            with torch.no_grad():
                Q, d = None, None  # Q[k] [F, width] / d[k] [width], None = zero
                for l, (wl, wr, b, _, _) in enumerate(maps):
                    scale, shift = affines[l] if l < last else (None, None)
                    ml = wl.t() if scale is None else wl.t() * scale[None, :]  # W_l^T S  [in, out]
                    mr = None if wr is None else (wr.t() if scale is None else wr.t() * scale[None, :])
                    if Q is None:  # first layer: H = A X M_l + X M_r + 1 b'
                        Q, d = [mr, ml], [None, None]
                    else:
                        nq, nd = [None] * (len(Q) + 1), [None] * (len(Q) + 1)
                        for k in range(len(Q)):
                            for src, m, dst in ((Q, mr, k), (Q, ml, k + 1)):
                                if src[k] is not None and m is not None:
                                    nq[dst] = src[k] @ m if nq[dst] is None else nq[dst] + src[k] @ m
                            for m, dst in ((mr, k), (ml, k + 1)):
                                if d[k] is not None and m is not None:
                                    nd[dst] = d[k] @ m if nd[dst] is None else nd[dst] + d[k] @ m
                        Q, d = nq, nd
                    const = b if scale is None else b * scale + shift
                    d[0] = const if d[0] is None else d[0] + const
                pad = torch.nn.functional.pad
                have = [k for k, q in enumerate(Q) if q is not None]
                weight = torch.cat([pad(Q[k].t(), (0, 0, 0, Cp - C)) for k in have], dim=0).contiguous()  # [blocks * C', F]
                shifts = [None if v is None else pad(v.detach(), (0, Cp - C)).contiguous() for v in d]
                out = (weight, have, shifts, C, Cp, maps[0][3], maps[0][4])
    self._collapsed = (key, out)
    return out


def _run_collapsed(self, x, edge_index, ce):
    """The eval forward (no_grad) through _collapsed_operands: one product at (blocks x) the class width, one row gather per
    layer with the lower-order block as its additive operand, the masked cross-entropy (one or two masks) taken in the last
    gather. None where the collapsed form does not apply."""
    from .. import ops
    from ..graph import get_graph
    operands = self._collapsed_operands()
    if operands is None:
        return None
    weight, have, shifts, C, Cp, kind, loops_mode = operands
    graph = get_graph(edge_index, x.size(0), loops_mode)
    if getattr(graph, "is_distributed", False) or not ops.rows_epilogue_ok(graph, Cp, x, None if ce is None else ce[0]):
        return None
    z = ops.linear(x, weight)  # [N, blocks * C']: over the non-zeros of bag-of-words features (ops.prepare_features)
    block = {k: z[:, j * Cp:(j + 1) * Cp] for j, k in enumerate(have)}
    w, rs = ops._kind_weights(graph, kind)
    top = len(shifts) - 1  # = number of layers: the highest power of A
    u = block[top]
    for k in range(top - 1, 0, -1):
        u = ops.spmm_raw(graph.fwd, w, rs, u, y=block.get(k), a=1.0, b=1.0, bias=shifts[k], kind=f"{kind}_fwd")
    if ce is None:
        return ops.spmm_raw(graph.fwd, w, rs, u, y=block.get(0), a=1.0, b=1.0, bias=shifts[0], kind=f"{kind}_fwd")[:, :C]
    labels, mask = ce
    _, stats = ops.spmm_epilogue_raw(graph.fwd, w, rs, u, y=block.get(0), a=1.0, b=1.0, bias=shifts[0],
                                     ce=(labels, mask, None), n_classes=C, kind=f"{kind}_fwd")
    return None, stats


ConvStack._collapsed_operands = _collapsed_operands
ConvStack._run_collapsed = _run_collapsed


def _eval_operands(self):
    """Per conv layer the operands of its eval forward — W'^T, b', Wr'^T with the eval-mode BatchNorm behind the layer
    folded in (rgbx_fold_bn_linear_f32, one launch per layer) — or None for layers without that form. Made once per
    model state and shared by the val and the test forward of an epoch: keyed by the training forwards taken (BatchNorm's
    kernels update the running statistics through raw pointers), by ops.weights_epoch() (a fused optimizer step does not
    move version counters; every optimizer step of the process moves that counter) plus every parameter's / buffer's
    version counter (load_state_dict and the like)."""
    tensors = list(self.parameters()) + list(self.buffers())
    from .. import ops
    key = (getattr(self, "_train_forwards", 0), ops.weights_epoch()) + tuple((t._version, t.data_ptr()) for t in tensors)
    cached = getattr(self, "_folded_eval", None)
    if cached is None or cached[0] != key:
        last = self.num_layers - 1
        out = []
        for i, conv in enumerate(self.convs):
            bn = self.bns[i] if i < last else None
            make = getattr(conv, "eval_operands", None)
            ok = make is not None and (bn is None or (hasattr(bn, "eval_affine") and bn.affine and bn.track_running_stats))
            out.append(make(bn) if ok else None)
        cached = (key, out)
        self._folded_eval = cached
    return cached[1]


ConvStack._eval_operands = _eval_operands


def _input_aggregate(self, x, edge_index):
    """The first conv's aggregate P x of the input features, kept across forwards and epochs (opt-in:
    `model.cache_input_aggregate = True`, experiment(cache_input_aggregate=True)). The reference recomputes it in every
    forward (models/gcn.py:27 inside the loop of itexperiments.py:417-473); with static features and a static graph it
    is the same matrix every time. Keyed by the feature tensor (address, version, shape) and the edge_index tensor;
    None where the first layer has no aggregate-first form (in > out, unsupported widths, features that take a
    gradient, partitioned graphs)."""
    if x.requires_grad or not x.is_cuda:
        return None
    key = (x.data_ptr(), x._version, tuple(x.shape), edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
    cached = getattr(self, "_agg0", None)
    if cached is None or cached[0] != key:
        make = getattr(self.convs[0], "aggregate_input", None)
        cached = (key, make(x, edge_index) if make is not None else None, x, edge_index)  # the tensors stay alive
        self._agg0 = cached
    return cached[1]


ConvStack._input_aggregate = _input_aggregate


def masked_ce_pair(model, fwd, y, mask_a, mask_b):
    """[2, 3] eval statistics of `model(**fwd)` under two masks from ONE forward (ConvStack.masked_ce_pair where the model
    has it, the logits otherwise): what share_eval_forward runs per epoch."""
    from .. import ops
    fn = getattr(model, "masked_ce_pair", None)
    if fn is not None and set(fwd) == {"x", "edge_index"} and fwd["x"].is_cuda:
        return fn(fwd["x"], fwd["edge_index"], y, mask_a, mask_b)
    emb = model(**fwd)["emb"]
    return torch.stack([ops.masked_ce_accuracy(emb, y, mask_a), ops.masked_ce_accuracy(emb, y, mask_b)])


def masked_ce(model, fwd, y, mask):
    """(loss, stats) of `model(**fwd)` on the masked rows, through model.masked_ce where the model has one (the conv
    stacks: loss inside the last conv's kernel) and from the logits otherwise."""
    from .. import ops
    fn = getattr(model, "masked_ce", None)
    if fn is not None and set(fwd) == {"x", "edge_index"} and fwd["x"].is_cuda:
        return fn(fwd["x"], fwd["edge_index"], y, mask)
    return ops.ce_from_logits(model(**fwd)["emb"], y, mask)
