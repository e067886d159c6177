"""GCN — reference models/gcn.py:5-31 (GCNConv stack with BatchNorm between layers)."""
from ..nn import GCNConv
from ._stack import ConvStack


class GCN(ConvStack):
    def __init__(self, num_layers, hidden_unit, input_dim, output_dim, dropout_rate):
        widths = [input_dim] + [hidden_unit] * (num_layers - 1) + [output_dim]
        super().__init__(num_layers, dropout_rate, widths, lambda i, a, b: GCNConv(a, b), hidden_unit)
