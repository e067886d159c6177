"""Shared skeleton of the reference's conv-stack models: (conv -> BatchNorm1d) x (L-1), conv,
log_softmax. No activation and no dropout are applied (the reference stores ``dropout_rate`` but
never uses it: models/gcn.py:15,25-31)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn import BatchNorm1d


def model_output(logits):
    """The reference's forward contract: {'out': log-probs, 'emb': logits} (models/gcn.py:31).
    'x' aliases 'out' because README.md:51 names the key 'x' while itexperiments.py:428 reads 'out'."""
    out = F.log_softmax(logits, dim=1)
    return {"out": out, "emb": logits, "x": out}


class ConvStack(nn.Module):
    """`widths` = [in, hid, ..., hid, out]; `make_conv(i, fan_in, fan_out)` builds layer i;
    `bn_width` = feature width after every non-final conv."""

    def __init__(self, num_layers, dropout_rate, widths, make_conv, bn_width):
        super().__init__()
        if num_layers < 2:
            raise ValueError("num_layers must be at least 2")
        self.num_layers = num_layers
        self.dropout_rate = dropout_rate
        self.convs = nn.ModuleList(make_conv(i, widths[i], widths[i + 1]) for i in range(num_layers))
        self.bns = nn.ModuleList(BatchNorm1d(bn_width) for _ in range(num_layers - 1))

    def forward(self, x, edge_index):
        last = self.num_layers - 1
        for conv, bn in zip(self.convs[:last], self.bns):
            fold = getattr(bn, "eval_affine", None) if getattr(conv, "folds_post_affine", False) else None
            affine = fold() if fold is not None and not torch.is_grad_enabled() else None
            if affine is not None:  # eval forward: BatchNorm's affine map folded into the conv's weights
                x = conv(x, edge_index, post_affine=affine)
            else:
                x = bn(conv(x, edge_index))
        return model_output(self.convs[last](x, edge_index))
