"""APPNPStack — reference models/appnp_stack.py:8-31: lin1 -> BatchNorm1d -> lin2 -> APPNP(K, alpha);
the propagation runs at width output_dim."""
import torch.nn as nn

from .. import ops
from ..graph import LOOPS_ADD_REMAINING, get_graph
from ..nn import APPNP, BatchNorm1d
from ._stack import model_output


class APPNPStack(nn.Module):
    def __init__(self, hidden_unit, input_dim, output_dim, K, alpha, dropout_rate):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.lin1 = nn.Linear(input_dim, hidden_unit)
        self.lin2 = nn.Linear(hidden_unit, output_dim)
        self.bn = BatchNorm1d(hidden_unit)
        self.conv = APPNP(K, alpha)

    def _features(self, x, edge_index):
        """lin2(bn(lin1(x))); on one GPU with lin2's rows zero-padded to a multiple of 4 (C = 7 -> 8): the K propagates
        then run on 16-byte rows without a pad copy of the [N, C] matrix (ops._pad4). Returns (h [N, n], n, graph)."""
        graph = get_graph(edge_index, x.size(0), LOOPS_ADD_REMAINING) if x.is_cuda else None
        h = ops.linear(x, self.lin1.weight, self.lin1.bias)
        if graph is not None and not getattr(graph, "is_distributed", False):
            weight, bias, n = ops.pad_rows4(self.lin2.weight, self.lin2.bias)
        else:
            weight, bias, n = self.lin2.weight, self.lin2.bias, self.lin2.out_features
        return ops.linear(self.bn(h), weight, bias), n, graph

    def forward(self, x, edge_index):
        h, n, _ = self._features(x, edge_index)
        out = self.conv(h, edge_index)
        C = self.lin2.out_features
        return model_output(out if n == C else out[:, :C])

    def masked_ce(self, x, edge_index, y, mask):
        """(loss, stats) of the model's masked cross-entropy (see ConvStack.masked_ce) with the loss taken inside APPNP's
        last propagate (ops.appnp_propagate_ce): the logits are never written. `mask` may be a pair (masked_ce_pair)."""
        h, n, graph = self._features(x, edge_index)
        if graph is not None and self.conv.K >= 1 and ops.rows_epilogue_ok(graph, n, h, y):
            return ops.appnp_propagate_ce(h, graph, self.conv.K, self.conv.alpha, self.lin2.out_features, y, mask)
        out = self.conv(h, edge_index)
        return ops.ce_from_logits(out[:, :self.lin2.out_features], y, mask)

    def masked_ce_pair(self, x, edge_index, y, mask_a, mask_b):
        """[2, 3] statistics of ONE eval forward under two masks (val and test, itexperiments.py:464-473)."""
        return self.masked_ce(x, edge_index, y, (mask_a, mask_b))[1]
