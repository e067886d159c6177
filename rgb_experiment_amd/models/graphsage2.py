"""GraphSAGE2 — reference models/graphsage2.py:7-33 (PyG SAGEConv: aggregate first, then Linear)."""
from ..nn import SAGEConv
from ._stack import ConvStack


class GraphSAGE2(ConvStack):
    def __init__(self, num_layers, hidden_unit, input_dim, output_dim, dropout_rate):
        widths = [input_dim] + [hidden_unit] * (num_layers - 1) + [output_dim]
        super().__init__(num_layers, dropout_rate, widths, lambda i, a, b: SAGEConv(a, b), hidden_unit)
