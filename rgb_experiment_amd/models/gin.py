"""GIN — reference models/gin.py:8-54: num_layers GINConv blocks (Linear-ReLU-Linear-ReLU-BatchNorm,
train_eps=True), then lin1 -> ReLU -> dropout -> lin2."""
import torch.nn as nn
import torch.nn.functional as F

from ..nn import BatchNorm1d, GINConv, Linear
from ._stack import model_output


def _block(fan_in, width):
    return nn.Sequential(Linear(fan_in, width), nn.ReLU(), Linear(width, width), nn.ReLU(),
                         BatchNorm1d(width))


class GIN(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_unit, num_layers, dropout_rate):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.conv1 = GINConv(_block(input_dim, hidden_unit), train_eps=True)
        self.convs = nn.ModuleList(GINConv(_block(hidden_unit, hidden_unit), train_eps=True)
                                   for _ in range(num_layers - 1))
        self.lin1 = Linear(hidden_unit, hidden_unit)
        self.lin2 = Linear(hidden_unit, output_dim)

    def forward(self, x, edge_index):
        x = self.conv1(x, edge_index)
        for conv in self.convs:
            x = conv(x, edge_index)
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout_rate, training=self.training)
        return model_output(self.lin2(x))
