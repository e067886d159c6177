"""SGC — reference models/sgc.py:6-14: a single SGConv(K, cached, add_self_loops)."""
import torch.nn as nn

from ..nn import SGConv
from ._stack import model_output


class SGC(nn.Module):
    def __init__(self, input_dim, output_dim, K, cached=True, add_self_loops=True):
        super().__init__()
        self.conv1 = SGConv(input_dim, output_dim, K=K, cached=cached, add_self_loops=add_self_loops)

    def forward(self, x, edge_index):
        return model_output(self.conv1(x, edge_index))
