"""Conv layers with the parameter names and semantics of the PyG layers the reference imports,
computing their ``propagate`` in HIP kernels (rgb_experiment_amd.ops).

  GCNConv    <- torch_geometric.nn.conv.GCNConv   (reference models/gcn.py:3,18-21)
  SAGEConv   <- torch_geometric.nn.conv.SAGEConv  (reference models/graphsage2.py:5,20-23)
  MySAGEConv <- my_SAGEConv                       (reference models/graphsage.py:36-62)
  GATConv    <- torch_geometric.nn.conv.GATConv   (reference models/gat.py:3,18-21)
  APPNP      <- torch_geometric.nn.conv.APPNP     (reference models/appnp_stack.py:3,22)

Dense X·W^T products go through hipBLASLt (forward, input gradient) and rgbx_gemm_tn_f32 (weight
gradient, split-K fp32 MFMA); everything indexed by edge_index goes through librgbx_hip.so.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..graph import LOOPS_ADD_REMAINING, LOOPS_KEEP, LOOPS_REMOVE_ADD, get_graph


def glorot_(t):
    """PyG's glorot: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)
    return t


def _forward_folded(conv, x, edge_index, operands, ce, loops_mode, kind, root):
    """Eval forward of a GCN / SAGE layer from prepared operands (W'^T, b', Wr'^T): ONE rgbx_fused_layer_f32 launch, no
    autograd node, no weight arithmetic on the way. Returns None when the fused kernel does not apply."""
    if operands is None or torch.is_grad_enabled() or not x.is_cuda:
        return None
    graph = get_graph(edge_index, x.size(0), loops_mode)
    if getattr(graph, "is_distributed", False) or not ops.fused_linear_ok(graph, conv.in_channels, conv.out_channels,
                                                                          root=root, x=x):
        return None
    wt, b, wtr = operands
    w = graph.w if kind == "gcn" else None
    rs = graph.inv_deg if kind == "mean" else None
    if ce is not None:
        y, mask = ce
        if not ops.fused_ce_ok(graph, conv.in_channels, conv.out_channels, root, x, y):
            return None
        _, _, stats = ops.fused_layer(x, wt, csr=graph.fwd, w=w, rs=rs, bias=b, x_root=x if root else None,
                                      wt_root=wtr if root else None, ce=(y, mask, None), kind=f"{kind}_linear_fwd")
        return None, stats  # an eval forward is read through its statistics; the mean loss is stats[0] / stats[1]
    out, _, _ = ops.fused_layer(x, wt, csr=graph.fwd, w=w, rs=rs, bias=b, x_root=x if root else None,
                                wt_root=wtr if root else None, kind=f"{kind}_linear_fwd")
    return out


def _aggregate_input(x, edge_index, loops_mode, kind):
    """P x of the static input features (no autograd), or None where the cached-aggregate route does not apply."""
    graph = get_graph(edge_index, x.size(0), loops_mode)
    if getattr(graph, "is_distributed", False) or not x.is_cuda or x.requires_grad:
        return None
    with torch.no_grad():
        return ops.propagate_gcn(x, graph) if kind == "gcn" else ops.propagate_mean(x, graph)


def _folded_from_aggregate(z, x, operands, root):
    """Eval forward from the kept aggregate and prepared operands (W'^T, b', Wr'^T): one DENSE launch."""
    wt, b, wtr = operands
    out, _, _ = ops.fused_layer(z, wt, bias=b, x_root=x if root else None, wt_root=wtr if root else None,
                                kind="cached_aggregate_linear_fwd")
    return out


class GCNConv(nn.Module):
    """out = A_hat (x W^T) + b, A_hat = D^-1/2 (A ∪ I) D^-1/2 with in-degree over the target index
    (gcn_norm restated at reference models/dagnn.py:12-31; message norm*x_j at dagnn.py:57-59).
    Parameters: ``lin.weight`` [out, in] (glorot, no bias), ``bias`` [out] (zeros)."""

    folds_post_affine = True  # forward(..., post_affine=(scale, shift)): see models/_stack.py
    emits_colsums = True      # forward(..., want_colsums=True): the output may carry its column sums (ops.COLSUMS)

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        nn.init.zeros_(self.bias)

    accepts_ce = True         # forward(..., ce=(y, mask)): the model's last layer may take the loss into its kernel
    accepts_ce_pair = True    # ... and mask may be (mask_a, mask_b): (None, [2, 3] statistics) of one eval forward

    def forward(self, x, edge_index, post_affine=None, want_colsums=False, ce=None):
        """`post_affine` = (scale, shift) of an eval-mode BatchNorm that follows this layer (no_grad only): a
        per-column affine map of a linear layer's output is the same layer with rows of W and b rescaled, so the
        normalisation costs two [out]-sized vector ops instead of a pass over [N, out].
        `want_colsums`: a training-mode BatchNorm follows; where the fused kernel runs, the output carries the column
        sums that BatchNorm needs (attribute ops.COLSUMS), taken from the MFMA tiles instead of a pass over it.
        `ce` = (y, mask): this is the model's last layer and the caller wants the masked cross-entropy of its logits,
        not the logits: returns (loss, stats [nll sum, selected rows, correct]); where the fused kernel runs the loss
        is taken from the output tiles and the logits are never written (ops.propagate_linear_ce)."""
        if ce is not None:
            return self._ce(x, edge_index, ce, None, None)
        weight, bias = self.lin.weight, self.bias
        if post_affine is not None:
            scale, shift = post_affine
            weight, bias = weight * scale[:, None], bias * scale + shift
        return self._conv(x, edge_index, weight, bias, want_colsums)

    def eval_operands(self, bn=None):
        """(W'^T, b', None) of this layer for an eval forward, the eval-mode BatchNorm `bn` behind it folded in."""
        return ops.fold_bn_linear(self.lin.weight, self.bias, bn=bn)

    # opt-in cache of the static input features' aggregate (models/_stack.ConvStack.cache_input_aggregate)
    def aggregate_input(self, x, edge_index):
        if self.in_channels > self.out_channels or not ops.aggregate_linear_ok(self.in_channels, self.out_channels):
            return None
        return _aggregate_input(x, edge_index, LOOPS_ADD_REMAINING, "gcn")

    def forward_from_aggregate(self, z, x, want_colsums=False, folded=None):
        if folded is not None:
            return _folded_from_aggregate(z, x, folded, False)
        return ops.aggregate_linear(z, self.lin.weight, self.bias, want_colsums=want_colsums)

    def forward_folded(self, x, edge_index, operands, ce=None):
        """Eval forward (no_grad) from prepared operands (eval_operands; models/_stack.ConvStack keeps them per
        parameter state): one fused launch. `ce` = (y, mask): returns (None, stats). None when this layer / graph does
        not take the fused kernel (the caller then runs the ordinary forward)."""
        return _forward_folded(self, x, edge_index, operands, ce, LOOPS_ADD_REMAINING, "gcn", False)

    def _ce(self, x, edge_index, ce, bn, colsums):
        y, mask = ce
        graph = get_graph(edge_index, x.size(0), LOOPS_ADD_REMAINING)
        hand_over = bn is not None and bn.folds_into_next_layer(x)
        if ops.fused_ce_ok(graph, self.in_channels, self.out_channels, False, x, y) and (bn is None or hand_over):
            return ops.propagate_linear_ce(x, graph, "gcn", self.lin.weight, self.bias, None, y, mask, bn=bn,
                                           colsums=colsums)
        if bn is not None:
            x = bn(x, colsums=colsums)
        weight, bias, n = ops.pad_rows4(self.lin.weight, self.bias)
        if ops.rows_epilogue_ok(graph, n, x, y) and not ops.fused_linear_ok(graph, self.in_channels, self.out_channels, x=x):
            # transform first (the reference's shapes: hidden 64 -> C = 7, initial_params.py:25), then aggregate the
            # [N, 8] rows with the loss taken in the gather kernel: no logits, no log-softmax / NLL / arg-max passes
            return ops.propagate_rows_ce(ops.linear(x, weight), graph, "gcn", n, self.out_channels, y, mask, bias=bias)
        return ops.ce_from_logits(self.forward(x, edge_index), y, mask)

    def forward_after_bn(self, x, edge_index, bn, colsums=None, want_colsums=False, ce=None):
        """self(bn(x), edge_index) for the BatchNorm1d in front of this layer. In a training forward on one GPU the
        normalised matrix is not written: the fused kernel gathers the raw rows and applies BatchNorm's affine map to
        the aggregate (ops.bn_propagate_linear). `colsums`: the column sums of x if its producer took them."""
        if ce is not None:
            return self._ce(x, edge_index, ce, bn, colsums)
        graph = get_graph(edge_index, x.size(0), LOOPS_ADD_REMAINING)
        if (bn.folds_into_next_layer(x) and not getattr(graph, "is_distributed", False)
                and ops.fused_linear_ok(graph, self.in_channels, self.out_channels, x=x)):
            return ops.bn_propagate_linear(x, bn, graph, "gcn", self.lin.weight, self.bias, colsums=colsums,
                                           want_colsums=want_colsums)
        return self.forward(bn(x, colsums=colsums), edge_index, want_colsums=want_colsums)

    def _conv(self, x, edge_index, weight, bias, want_colsums=False):
        graph = get_graph(edge_index, x.size(0), LOOPS_ADD_REMAINING)
        if ops.fused_linear_ok(graph, self.in_channels, self.out_channels, x=x):
            # A_hat (x W^T) + b = (A_hat x) W^T + b in one kernel: the aggregate stays in LDS and the GEMM
            # runs on the MFMA units underneath the gather (ops._PropagateLinear)
            return ops.propagate_linear(x, graph, "gcn", weight, bias, want_colsums=want_colsums)
        if not x.requires_grad and self.in_channels <= self.out_channels:
            # Input layer (and every layer under no_grad): A_hat (x W^T) = (A_hat x) W^T. Aggregating first
            # costs the same forward (in <= out) and makes dW = dy^T (A_hat x) a plain weight-gradient GEMM: no
            # gradient has to travel back through A_hat^T, because x needs none (one transposed SpMM less per
            # step). On a partitioned graph the aggregated tensor is then the static feature matrix, whose
            # boundary rows are resident (dist.DistGraph.pin_resident): no exchange either.
            return ops.linear(ops.propagate_gcn(x, graph), weight, bias)
        if want_colsums and ops.rows_epilogue_ok(graph, self.out_channels, x):
            # transform first; the BatchNorm behind the layer gets its column sums from the gather kernel
            return ops.propagate_rows(ops.linear(x, weight), graph, "gcn", bias=bias, want_colsums=True)
        return ops.propagate_gcn(ops.linear(x, weight), graph, bias=bias)


class SAGEConv(nn.Module):
    """out = lin_l(mean_{j in N(i)} x_j) + lin_r(x_i); no self-loops; nodes without in-edges
    aggregate 0 [PyG SAGEConv defaults: aggr='mean', root_weight=True, lin_l bias, lin_r no bias]."""

    folds_post_affine = True  # forward(..., post_affine=(scale, shift)): see models/_stack.py
    emits_colsums = True      # see GCNConv

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    accepts_ce = True         # see GCNConv
    accepts_ce_pair = True

    def eval_operands(self, bn=None):
        return ops.fold_bn_linear(self.lin_l.weight, self.lin_l.bias, root_weight=self.lin_r.weight, bn=bn)

    def aggregate_input(self, x, edge_index):
        if self.in_channels > self.out_channels or not ops.aggregate_linear_ok(self.in_channels, self.out_channels, True):
            return None
        return _aggregate_input(x, edge_index, LOOPS_KEEP, "mean")

    def forward_from_aggregate(self, z, x, want_colsums=False, folded=None):
        if folded is not None:
            return _folded_from_aggregate(z, x, folded, True)
        return ops.aggregate_linear(z, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, x, want_colsums)

    def forward_folded(self, x, edge_index, operands, ce=None):
        """See GCNConv.forward_folded."""
        return _forward_folded(self, x, edge_index, operands, ce, LOOPS_KEEP, "mean", True)

    def _ce(self, x, edge_index, ce, bn, colsums):
        y, mask = ce
        graph = get_graph(edge_index, x.size(0), LOOPS_KEEP)
        hand_over = bn is not None and bn.folds_into_next_layer(x)
        if ops.fused_ce_ok(graph, self.in_channels, self.out_channels, True, x, y) and (bn is None or hand_over):
            return ops.propagate_linear_ce(x, graph, "mean", self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, y,
                                           mask, bn=bn, colsums=colsums)
        if bn is not None:
            x = bn(x, colsums=colsums)
        if (ops.rows_epilogue_ok(graph, (self.out_channels + 3) // 4 * 4, x, y)
                and not ops.fused_linear_ok(graph, self.in_channels, self.out_channels, root=True, x=x)):
            h, n = self._transform_first(x, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight)
            return ops.propagate_rows_ce(h, graph, "mean", n, self.out_channels, y, mask)
        return ops.ce_from_logits(self.forward(x, edge_index), y, mask)

    @staticmethod
    def _transform_first(x, w_l, b_l, w_r):
        """h = x [W_l; W_r]^T + [0; b_l] as ONE product ([N, 2 n], n = out rounded up to a multiple of 4): the left half is
        what the mean runs over, the right half the root term plus lin_l's bias (which PyG adds after the aggregation:
        a node without in-edges gets b_l + W_r x_i). The wide input (F = 1433 on Cora) is read once for both Linears."""
        w_l, b_l, n = ops.pad_rows4(w_l, b_l)
        w_r = ops.pad_rows4(w_r)[0]
        return ops.linear(x, torch.cat([w_l, w_r]), torch.cat([torch.zeros_like(b_l), b_l])), n

    def forward_after_bn(self, x, edge_index, bn, colsums=None, want_colsums=False, ce=None):
        """See GCNConv.forward_after_bn; the root term lin_r(bn(x)_i) gets the affine map as its rows are loaded."""
        if ce is not None:
            return self._ce(x, edge_index, ce, bn, colsums)
        graph = get_graph(edge_index, x.size(0), LOOPS_KEEP)
        if (bn.folds_into_next_layer(x) and not getattr(graph, "is_distributed", False)
                and ops.fused_linear_ok(graph, self.in_channels, self.out_channels, root=True, x=x)):
            return ops.bn_propagate_linear(x, bn, graph, "mean", self.lin_l.weight, self.lin_l.bias,
                                           root_weight=self.lin_r.weight, colsums=colsums, want_colsums=want_colsums)
        return self.forward(bn(x, colsums=colsums), edge_index, want_colsums=want_colsums)

    def forward(self, x, edge_index, post_affine=None, want_colsums=False, ce=None):
        if ce is not None:
            return self._ce(x, edge_index, ce, None, None)
        w_l, b_l, w_r = self.lin_l.weight, self.lin_l.bias, self.lin_r.weight
        if post_affine is not None:  # see GCNConv.forward
            scale, shift = post_affine
            w_l, b_l, w_r = w_l * scale[:, None], b_l * scale + shift, w_r * scale[:, None]
        graph = get_graph(edge_index, x.size(0), LOOPS_KEEP)
        if ops.fused_linear_ok(graph, self.in_channels, self.out_channels, root=True, x=x):
            # lin_l(mean_j x_j) + lin_r(x_i) in one kernel: both products accumulate in the same MFMA tile
            return ops.propagate_linear(x, graph, "mean", w_l, b_l, root_weight=w_r, want_colsums=want_colsums)
        if self.in_channels > self.out_channels and x.is_cuda and not getattr(graph, "is_distributed", False):
            # in > out (the reference's defaults: F -> 64 -> C; graphsage2 at F = 1433 is the row its README marks OOM,
            # README.md:74): mean_j(x_j) W_l^T = mean_j(x_j W_l^T) — transform first, gather at the OUTPUT width
            h, n = self._transform_first(x, w_l, b_l, w_r)
            out = ops.propagate_rows(h, graph, "mean", n=n, want_colsums=want_colsums and n == self.out_channels)
            return out if n == self.out_channels else out[:, :self.out_channels]
        x_r = ops.linear(ops.target_rows(x, graph), w_r)  # the targets' own rows (all of x except on a dist.ReplicaGraph)
        if ops.fused_linear_ok(graph, self.in_channels, self.out_channels, x=x):
            return ops.propagate_linear(x, graph, "mean", w_l, b_l) + x_r
        if self.in_channels > self.out_channels:
            return ops.propagate_mean(ops.linear(x, w_l), graph) + b_l + x_r
        agg = ops.propagate_mean(x, graph)
        return ops.linear(agg, w_l, b_l) + x_r


class MySAGEConv(nn.Module):
    """reference models/graphsage.py:36-62: x_l = lin_l(x), x_r = lin_r(x) (both with bias),
    remove_self_loops + add_self_loops, mean over N(i) ∪ {i} of x_l, then += x_r."""

    folds_post_affine = True  # forward(..., post_affine=(scale, shift)): see models/_stack.py
    emits_colsums = True      # see GCNConv

    def __init__(self, in_channels, out_channels, add_self_loops=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.add_self_loops = add_self_loops
        self.lin_l = nn.Linear(in_channels, out_channels)
        self.lin_r = nn.Linear(in_channels, out_channels)

    accepts_ce = True         # see GCNConv
    accepts_ce_pair = True

    def eval_operands(self, bn=None):
        if not self.add_self_loops:
            return None
        return ops.fold_bn_linear(self.lin_l.weight, self.lin_l.bias, self.lin_r.bias, root_weight=self.lin_r.weight,
                                  bn=bn)

    def aggregate_input(self, x, edge_index):
        if (not self.add_self_loops or self.in_channels > self.out_channels
                or not ops.aggregate_linear_ok(self.in_channels, self.out_channels, True)):
            return None
        return _aggregate_input(x, edge_index, LOOPS_REMOVE_ADD, "mean")

    def forward_from_aggregate(self, z, x, want_colsums=False, folded=None):
        if folded is not None:
            return _folded_from_aggregate(z, x, folded, True)
        return ops.aggregate_linear(z, self.lin_l.weight, self.lin_l.bias + self.lin_r.bias, self.lin_r.weight, x,
                                    want_colsums)

    def forward_folded(self, x, edge_index, operands, ce=None):
        """See GCNConv.forward_folded (mean over N(i) + {i}: the weights of a row sum to 1, so both biases and the
        BatchNorm shift ride in the kernel's bias)."""
        return _forward_folded(self, x, edge_index, operands, ce, LOOPS_REMOVE_ADD, "mean", True)

    def _ce(self, x, edge_index, ce, bn, colsums):
        y, mask = ce
        if self.add_self_loops:
            graph = get_graph(edge_index, x.size(0), LOOPS_REMOVE_ADD)
            hand_over = bn is not None and bn.folds_into_next_layer(x)
            if ops.fused_ce_ok(graph, self.in_channels, self.out_channels, True, x, y) and (bn is None or hand_over):
                return ops.propagate_linear_ce(x, graph, "mean", self.lin_l.weight, self.lin_l.bias + self.lin_r.bias,
                                               self.lin_r.weight, y, mask, bn=bn, colsums=colsums)
        if bn is not None:
            x = bn(x, colsums=colsums)
        mode = LOOPS_REMOVE_ADD if self.add_self_loops else LOOPS_KEEP
        graph = get_graph(edge_index, x.size(0), mode)
        fused = self.add_self_loops and ops.fused_linear_ok(graph, self.in_channels, self.out_channels, root=True, x=x)
        if ops.rows_epilogue_ok(graph, (self.out_channels + 3) // 4 * 4, x, y) and not fused:
            h, n = self._transform_both(x, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, self.lin_r.bias)
            return ops.propagate_rows_ce(h, graph, "mean", n, self.out_channels, y, mask)
        return ops.ce_from_logits(self.forward(x, edge_index), y, mask)

    @staticmethod
    def _transform_both(x, w_l, b_l, w_r, b_r):
        """h = [lin_l(x), lin_r(x)] as ONE product (models/graphsage.py:49-50 runs two Linears over the same x): [N, 2 n],
        n = out rounded up to a multiple of 4. b_l stays inside the mean, as in the reference (a row without entries —
        add_self_loops=False and no in-edge — gets no b_l)."""
        w_l, b_l, n = ops.pad_rows4(w_l, b_l)
        w_r, b_r, _ = ops.pad_rows4(w_r, b_r)
        return ops.linear(x, torch.cat([w_l, w_r]), torch.cat([b_l, b_r])), n

    def forward_after_bn(self, x, edge_index, bn, colsums=None, want_colsums=False, ce=None):
        """See GCNConv.forward_after_bn."""
        if ce is not None:
            return self._ce(x, edge_index, ce, bn, colsums)
        if self.add_self_loops:
            graph = get_graph(edge_index, x.size(0), LOOPS_REMOVE_ADD)
            if (bn.folds_into_next_layer(x) and not getattr(graph, "is_distributed", False)
                    and ops.fused_linear_ok(graph, self.in_channels, self.out_channels, root=True, x=x)):
                return ops.bn_propagate_linear(x, bn, graph, "mean", self.lin_l.weight,
                                               self.lin_l.bias + self.lin_r.bias, root_weight=self.lin_r.weight,
                                               colsums=colsums, want_colsums=want_colsums)
        return self.forward(bn(x, colsums=colsums), edge_index, want_colsums=want_colsums)

    def forward(self, x, edge_index, post_affine=None, want_colsums=False, ce=None):
        if ce is not None:
            return self._ce(x, edge_index, ce, None, None)
        w_l, b_l, w_r, b_r = self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, self.lin_r.bias
        if post_affine is not None and self.add_self_loops:  # see GCNConv.forward; the mean weights sum to 1
            scale, shift = post_affine
            w_l, b_l = w_l * scale[:, None], b_l * scale
            w_r, b_r = w_r * scale[:, None], b_r * scale + shift
        out = self._conv(x, edge_index, w_l, b_l, w_r, b_r, want_colsums)
        if post_affine is not None and not self.add_self_loops:
            out = out * post_affine[0] + post_affine[1]
        return out

    def _conv(self, x, edge_index, w_l, b_l, w_r, b_r, want_colsums=False):
        mode = LOOPS_REMOVE_ADD if self.add_self_loops else LOOPS_KEEP
        graph = get_graph(edge_index, x.size(0), mode)
        if self.add_self_loops and ops.fused_linear_ok(graph, self.in_channels, self.out_channels, root=True, x=x):
            # mean_j(lin_l(x_j)) + lin_r(x_i) = (mean_j x_j) Wl^T + x_i Wr^T + (b_l + b_r), one kernel
            return ops.propagate_linear(x, graph, "mean", w_l, b_l + b_r, root_weight=w_r, want_colsums=want_colsums)
        fused_left = self.add_self_loops and ops.fused_linear_ok(graph, self.in_channels, self.out_channels, x=x)
        input_layer = not x.requires_grad and self.add_self_loops and self.in_channels <= self.out_channels
        if x.is_cuda and not getattr(graph, "is_distributed", False) and not fused_left and not input_layer:
            # transform first, as the reference writes the layer (in > out is its default shape: F -> 64 -> C)
            return self._conv_rows(x, graph, w_l, b_l, w_r, b_r, want_colsums)
        x_r = ops.linear(ops.target_rows(x, graph), w_r, b_r)
        if fused_left:
            return ops.propagate_linear(x, graph, "mean", w_l, b_l) + x_r
        if input_layer:
            # Input layer (see GCNConv.forward): with the self-loop every row's mean weights
            # sum to 1, so mean_j(W x_j + b) = W mean_j(x_j) + b exactly; aggregating first removes the
            # transposed SpMM from this layer's backward.
            return ops.linear(ops.propagate_mean(x, graph), w_l, b_l) + x_r
        x_l = ops.linear(x, w_l, b_l)
        return ops.propagate_mean(x_l, graph) + x_r

    def _conv_rows(self, x, graph, w_l, b_l, w_r, b_r, want_colsums):
        """The layer as the reference writes it — transform, then the mean, then += x_r (graphsage.py:49-60) — on one
        product and one gather: the add is the gather kernel's additive operand."""
        h, n = self._transform_both(x, w_l, b_l, w_r, b_r)
        out = ops.propagate_rows(h, graph, "mean", n=n, want_colsums=want_colsums and n == self.out_channels)
        return out if n == self.out_channels else out[:, :self.out_channels]


class GATConv(nn.Module):
    """h = x W^T viewed [N,H,C]; e_ij = LeakyReLU(<h_j,att_src> + <h_i,att_dst>, 0.2); softmax over
    the in-edges of i (self-loops removed then re-added); out_i = sum_j alpha_ij h_j; heads
    concatenated (or averaged when concat=False); + bias [PyG GATConv defaults; dropout 0 as the
    reference never sets it, models/gat.py:18-21]."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.heads, self.concat, self.negative_slope = heads, concat, negative_slope
        self.lin_src = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.lin_dst = self.lin_src  # shared, as PyG does for a single feature matrix
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin_src.weight)
        glorot_(self.att_src)
        glorot_(self.att_dst)
        nn.init.zeros_(self.bias)

    folds_post_affine = True  # forward(..., post_affine=(scale, shift)): see models/_stack.py
    accepts_ce = True  # forward(..., ce=(y, mask)) returns (loss, stats): see models/_stack.py

    @staticmethod
    def kernel_channels(C):
        """Channels per head the attention kernels run with. A wave holds one head's channels on at most 64 lanes of
        4 / 2 / 1 floats (csrc/gat.hip make_layout: C % 4 == 0 up to 256, C % 2 == 0 up to 128, any C up to 64); other
        widths — 130 classes on the single-head output layer, an odd 67 — are padded per head to the next multiple of 4
        with zero weight rows, zero attention entries and zero bias, which changes neither a score nor a kept column
        (PyG's GATConv has no such bound)."""
        if C <= 64 or (C % 2 == 0 and C <= 128) or (C % 4 == 0 and C <= 256):
            return C
        Cp = (C + 3) // 4 * 4
        if Cp > 256:
            raise NotImplementedError(f"GATConv: {C} channels per head; the attention kernels take at most 256")
        return Cp

    def forward(self, x, edge_index, post_affine=None, ce=None):
        """`post_affine` = (scale, shift) of an eval-mode BatchNorm that follows this layer (no_grad only): applied in
        the aggregation kernel's store, out = aggregate * scale + (bias * scale + shift), when the bias rides there
        too; otherwise after the layer. `ce` = (y, mask): the layer is the model's last; returns (loss, stats) of the
        masked cross-entropy of its output instead of the output."""
        H, C = self.heads, self.out_channels
        Cp = self.kernel_channels(C)
        if Cp == C:
            return self._forward(x, edge_index, self.lin_src.weight, self.att_src, self.att_dst, self.bias, C, post_affine, ce)
        pad = torch.nn.functional.pad
        in_kernel = self.concat or H == 1
        weight = pad(self.lin_src.weight.view(H, C, -1), (0, 0, 0, Cp - C)).reshape(H * Cp, -1)
        bias = pad(self.bias.view(H, C), (0, Cp - C)).reshape(-1) if in_kernel else pad(self.bias, (0, Cp - C))
        out = self._forward(x, edge_index, weight, pad(self.att_src, (0, Cp - C)), pad(self.att_dst, (0, Cp - C)), bias, Cp,
                            None, None)
        out = out.view(-1, H, Cp)[:, :, :C].reshape(-1, H * C) if (self.concat and H > 1) else out[:, :C]
        if post_affine is not None:
            out = out * post_affine[0] + post_affine[1]
        return out if ce is None else ops.ce_from_logits(out, ce[0], ce[1])

    def _forward(self, x, edge_index, weight, att_src, att_dst, bias, C, post_affine, ce):
        H = self.heads
        graph = get_graph(edge_index, x.size(0), LOOPS_REMOVE_ADD)
        if (H == 1 and post_affine is None
                and ops.gat_linear_ok(graph, self.in_channels, C, x, None if ce is None else ce[0])):
            # one head: sum_j alpha_ij (W x_j) = W sum_j alpha_ij x_j — scores from x, the coefficients as a per-edge
            # vector, then aggregation + transform (+ loss) in ONE launch of the fused kernel; no h = x W^T product
            return ops.gat_attend_linear(x, weight, att_src, att_dst, graph, self.negative_slope, bias=bias, ce=ce)
        if ce is not None:
            return ops.ce_from_logits(self._forward(x, edge_index, weight, att_src, att_dst, bias, C, post_affine, None),
                                      ce[0], ce[1])
        # the bias rides in the aggregation kernel's store when it applies to the stored row as is
        # (concatenated heads, or a single head, whose "mean over heads" is the identity)
        in_kernel = self.concat or H == 1
        dist_resident = getattr(graph, "is_distributed", False) and graph.is_resident(x)
        if dist_resident:
            out = graph.gat(x, att_src, att_dst, H, C, self.negative_slope, weight=weight)
            out = out + bias if in_kernel else out.view(-1, H, C).mean(dim=1) + bias
        else:
            h = ops.linear(x, weight)
            if in_kernel and post_affine is not None and not getattr(graph, "is_distributed", False):
                scale, shift = post_affine
                return ops.gat_attend(h, att_src, att_dst, graph, H, C, self.negative_slope,
                                      bias=bias * scale + shift, out_scale=scale)
            out = ops.gat_attend(h, att_src, att_dst, graph, H, C, self.negative_slope,
                                 bias=bias if in_kernel else None)
            if not in_kernel:
                out = out.view(-1, H, C).mean(dim=1) + bias
        if post_affine is not None:
            out = out * post_affine[0] + post_affine[1]
        return out


class APPNP(nn.Module):
    """z^0 = x; z^{k+1} = (1-alpha) A_hat z^k + alpha x, K times (in-repo twin of the recurrence:
    reference models/pta.py:79-84); gcn_norm computed once per graph."""

    def __init__(self, K, alpha):
        super().__init__()
        self.K, self.alpha = K, alpha

    def forward(self, x, edge_index):
        graph = get_graph(edge_index, x.size(0), LOOPS_ADD_REMAINING)
        return ops.appnp_propagate(x, graph, self.K, self.alpha)


class SGConv(nn.Module):
    """x' = lin(A_hat^K x) with gcn_norm (self-loops added unless add_self_loops=False) and the
    propagated features cached after the first call when cached=True [PyG SGConv, as built at reference
    models/sgc.py:9-10]. The K propagates run as one rgbx_appnp_f32 call with alpha = 0."""

    def __init__(self, in_channels, out_channels, K=1, cached=False, add_self_loops=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.cached, self.add_self_loops = cached, add_self_loops
        self.lin = nn.Linear(in_channels, out_channels, bias=bias)
        self._cached_x = None

    def forward(self, x, edge_index):
        h = self._cached_x
        if h is None:
            mode = LOOPS_ADD_REMAINING if self.add_self_loops else LOOPS_KEEP
            graph = get_graph(edge_index, x.size(0), mode)
            if not self.cached and self.in_channels > self.out_channels:
                # nothing is kept between calls, so A_hat^K (x W^T) = (A_hat^K x) W^T runs the K gathers at the
                # OUTPUT width (F = 1433 -> C = 7 on Cora: 1/180 of the bytes)
                out = ops.appnp_propagate(ops.linear(x, self.lin.weight), graph, self.K, 0.0)
                return out if self.lin.bias is None else out + self.lin.bias
            h = ops.appnp_propagate(x, graph, self.K, 0.0)
            if self.cached:
                self._cached_x = h if h.requires_grad else h.detach()
        return ops.linear(h, self.lin.weight, self.lin.bias)


class GINConv(nn.Module):
    """out = nn((1 + eps) * x_i + sum_{j in N(i)} x_j), edges as given, eps learnable when train_eps
    [PyG GINConv, as built at reference models/gin.py:14-34]."""

    def __init__(self, nn_module, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))

    def forward(self, x, edge_index):
        graph = get_graph(edge_index, x.size(0), LOOPS_KEEP)
        first = self.nn[0] if isinstance(self.nn, nn.Sequential) and len(self.nn) else None
        if (isinstance(first, nn.Linear) and x.is_cuda
                and ops.fused_linear_ok(graph, first.in_features, first.out_features, root=True, x=x)):
            # nn's first Linear applied to (sum_j x_j + (1 + eps) x_i) = (sum_j x_j) W^T + x_i ((1 + eps) W)^T + b:
            # aggregation, the root term and that Linear in one launch (rgbx_spmm_linear_f32), no [N, in] sum
            # written, no scale / add passes; eps gets its gradient through the root operand
            h = ops.propagate_linear(x, graph, "sum", first.weight, first.bias,
                                     root_weight=(1 + self.eps) * first.weight)
            for layer in list(self.nn)[1:]:
                h = layer(h)
            return h
        if isinstance(first, nn.Linear) and first.in_features > first.out_features:
            # (sum_j x_j + (1 + eps) x_i) W^T + b = sum_j (x_j W^T) + (1 + eps) (x_i W^T) + b: transform first, gather at
            # the output width of nn's first Linear (in > out: the reference's first block, F -> 64, models/gin.py:14-21)
            h = ops.linear(x, first.weight)
            h = ops.propagate_sum(h, graph) + (1 + self.eps) * h
            if first.bias is not None:
                h = h + first.bias
            for layer in list(self.nn)[1:]:
                h = layer(h)
            return h
        return self.nn(ops.propagate_sum(x, graph) + (1 + self.eps) * x)
