"""MLP — reference models/mlp.py:4-33; no graph input. Kept for plumbing tests of experiment().
The reference builds one more BatchNorm1d (over output_dim) than it applies (mlp.py:24); it is kept
so that state_dict keys match."""
import torch.nn as nn

from ..nn import BatchNorm1d, Linear
from ._stack import model_output


class MLP(nn.Module):
    def __init__(self, num_layers, hidden_unit, input_dim, output_dim, dropout_rate):
        super().__init__()
        self.num_layers = num_layers
        self.dropout_rate = dropout_rate
        widths = [input_dim] + [hidden_unit] * (num_layers - 1) + [output_dim]
        self.lins = nn.ModuleList(Linear(widths[i], widths[i + 1]) for i in range(num_layers))
        self.bns = nn.ModuleList([BatchNorm1d(hidden_unit) for _ in range(num_layers - 1)] +
                                 [BatchNorm1d(output_dim)])

    def forward(self, x):
        for i in range(self.num_layers - 1):
            x = self.bns[i](self.lins[i](x))
        return model_output(self.lins[-1](x))
