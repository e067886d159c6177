"""GAT — reference models/gat.py:5-32: GATConv(in, hid, heads) ..., GATConv(hid*heads, out, 1,
concat=False), BatchNorm1d(hid*heads) between layers."""
from ..nn import GATConv
from ._stack import ConvStack


class GAT(ConvStack):
    def __init__(self, num_layers, hidden_unit, input_dim, output_dim, dropout_rate, heads):
        wide = hidden_unit * heads
        widths = [input_dim] + [wide] * (num_layers - 1) + [output_dim]

        def make(i, fan_in, fan_out):
            if i == num_layers - 1:
                return GATConv(fan_in, output_dim, 1, concat=False)
            return GATConv(fan_in, hidden_unit, heads)

        super().__init__(num_layers, dropout_rate, widths, make, wide)
        self.heads = heads
