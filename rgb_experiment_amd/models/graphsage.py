"""GraphSAGE — reference models/graphsage.py:6-32, built on the in-repo my_SAGEConv (:36-62)."""
from ..nn import MySAGEConv
from ._stack import ConvStack

my_SAGEConv = MySAGEConv  # the reference's class name


class GraphSAGE(ConvStack):
    def __init__(self, num_layers, hidden_unit, input_dim, output_dim, dropout_rate):
        widths = [input_dim] + [hidden_unit] * (num_layers - 1) + [output_dim]
        super().__init__(num_layers, dropout_rate, widths, lambda i, a, b: MySAGEConv(a, b), hidden_unit)
