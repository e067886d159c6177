"""PTA — reference models/pta.py:12-84: a 2-layer MLP trained against propagated soft labels, with
APPNP-style inference. The propagation (`inference`, and `label_propagation` in itexperiments) runs on
the HIP SpMM through `NormAdj`; the reference uses a torch sparse COO matmul (pta.py:83,
itexperiments.py:715)."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import LOOPS_KEEP, get_graph


class NormAdj:
    """D^-1/2 (A + I) D^-1/2 exactly as reference itexperiments.py:354-356,671-684 builds it: entry
    adj[src, dst] per edge (duplicates add up, an existing self-loop gets the extra +1 of the identity),
    D = row sums, and `adj @ y` aggregates INTO the source index. Held as a target-grouped CSR of the
    reversed, loop-augmented edge list."""

    def __init__(self, edge_index, num_nodes):
        loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
        # aggregate at edge_index[0] from edge_index[1]  ->  (gather-from, aggregate-into) = (dst, src)
        self.edge_index = torch.cat([edge_index.flip(0), loops.unsqueeze(0).repeat(2, 1)], dim=1).contiguous()
        self.num_nodes = num_nodes
        self.graph = get_graph(self.edge_index, num_nodes, LOOPS_KEEP)

    def matmul(self, y):
        return ops.propagate_gcn(y, self.graph)


class Linear(nn.Module):
    """dropout + x @ weight (+ bias), weight stored [in, out] (reference pta.py:12-37)."""

    def __init__(self, in_features, out_features, dropout, bias=False):
        super().__init__()
        self.dropout, self.in_features, self.out_features = dropout, in_features, out_features
        # torch.randn, as the reference allocates them (models/pta.py:19,21): the draws advance the global RNG before
        # kaiming_uniform_ does, so a seeded run (need_to_reappear / reappear_seed) starts from the reference's weights
        self.weight = nn.Parameter(torch.randn(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.randn(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, mode="fan_out", a=math.sqrt(5))
        if self.bias is not None:
            stdv = 1.0 / math.sqrt(self.weight.size(1))
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, x):
        x = F.dropout(x, self.dropout, training=self.training)
        out = torch.matmul(x, self.weight)
        return out if self.bias is None else out + self.bias


class PTA(nn.Module):
    def __init__(self, nfeat, nhid, nclass, dropout, epsilon, K, alpha, mode=2):
        super().__init__()
        self.Linear1 = Linear(nfeat, nhid, dropout, bias=True)
        self.Linear2 = Linear(nhid, nclass, dropout, bias=True)
        self.epsilon, self.mode, self.K, self.alpha, self.number_class = epsilon, mode, K, alpha, nclass

    def forward(self, x):
        return self.Linear2(torch.relu(self.Linear1(x)))

    def loss_function(self, y_hat, y_soft, epoch=0):
        """reference pta.py:62-77: PTS / PTD / PTA weighting in training, plain soft-label CE in eval."""
        logp = torch.log_softmax(y_hat, dim=-1)
        if self.training and self.mode in (1, 2):
            conf = torch.softmax(y_hat, dim=-1).detach()
            if self.mode == 2:
                conf = conf ** np.log(epoch / self.epsilon + 1)
            return -torch.sum(logp * (y_soft * conf)) / self.number_class
        return -torch.sum(logp * y_soft) / self.number_class

    def inference(self, h, adj):
        """reference pta.py:79-84: y <- (1-alpha) adj @ y + alpha * softmax(h), K times."""
        y0 = torch.softmax(h, dim=-1)
        y = y0
        for _ in range(self.K):
            y = (1 - self.alpha) * adj.matmul(y) + self.alpha * y0
        return y
