"""DAGNN — reference models/dagnn.py:34-86: MLP (lin1 -> ReLU -> lin2, dropout before each Linear),
then Prop: K gcn-normalised propagates, the K+1 hops stacked and mixed by a learned sigmoid gate."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import LOOPS_ADD_REMAINING, get_graph
from ..nn import Linear
from ._stack import model_output


class Prop(nn.Module):
    """reference dagnn.py:34-55: preds = [x, A_hat x, ..., A_hat^K x]; retain = sigmoid(proj(preds));
    out = sum_k retain_k * preds_k."""

    def __init__(self, num_classes, K):
        super().__init__()
        self.K = K
        self.proj = nn.Linear(num_classes, 1)

    def forward(self, x, edge_index):
        graph = get_graph(edge_index, x.size(0), LOOPS_ADD_REMAINING)
        out = ops.dagnn_prop(x, graph, self.K, self.proj.weight, self.proj.bias)  # hops mixed where the SpMM left them
        if out is not None:
            return out
        preds = [x]
        for _ in range(self.K):
            x = ops.propagate_gcn(x, graph)
            preds.append(x)
        pps = torch.stack(preds, dim=1)                      # [N, K+1, C]
        retain = torch.sigmoid(self.proj(pps).squeeze(-1))   # [N, K+1]
        return torch.matmul(retain.unsqueeze(1), pps).squeeze(1)

    def reset_parameters(self):
        self.proj.reset_parameters()


class DAGNN(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, K, dropout_rate):
        super().__init__()
        self.lin1 = Linear(input_dim, hidden_dim)
        self.lin2 = Linear(hidden_dim, output_dim)
        self.prop = Prop(output_dim, K)
        self.dropout_rate = dropout_rate

    def reset_parameters(self):
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()
        self.prop.reset_parameters()

    def forward(self, x, edge_index):
        x = ops.dropout(x, self.dropout_rate, self.training)  # features carried by their non-zeros stay on them
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=self.dropout_rate, training=self.training)
        x = self.lin2(x)
        return model_output(self.prop(x, edge_index))
