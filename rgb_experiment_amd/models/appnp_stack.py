"""APPNPStack — reference models/appnp_stack.py:8-31: lin1 -> BatchNorm1d -> lin2 -> APPNP(K, alpha);
the propagation runs at width output_dim."""
import torch.nn as nn

from .. import ops
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

    def forward(self, x, edge_index):
        h = ops.linear(x, self.lin1.weight, self.lin1.bias)
        h = ops.linear(self.bn(h), self.lin2.weight, self.lin2.bias)
        return model_output(self.conv(h, edge_index))
