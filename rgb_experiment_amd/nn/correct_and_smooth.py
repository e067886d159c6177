"""Correct & Smooth post-processing (reference itexperiments.py:514-534 ->
torch_geometric.nn.CorrectAndSmooth / LabelPropagation [PyG, un-vendored]). Both stages are K
repetitions of `out <- alpha * A_hat out + (1 - alpha) * y0` followed by a clamp, with
A_hat = D^-1/2 A D^-1/2 WITHOUT added self-loops; each repetition is one rgbx_spmm_csr_f32 launch
with the residual fused into the store (a = alpha, y = (1-alpha) y0, b = 1)."""
import torch
import torch.nn.functional as F

from .. import ops
from ..graph import LOOPS_KEEP, get_graph


class LabelPropagation:
    def __init__(self, num_layers, alpha):
        self.num_layers, self.alpha = num_layers, alpha

    @torch.no_grad()
    def __call__(self, y, edge_index, post_step=None):
        post_step = post_step or (lambda t: t.clamp_(0.0, 1.0))
        graph = get_graph(edge_index, y.size(0), LOOPS_KEEP)
        out = y.contiguous()
        res = ((1 - self.alpha) * out).contiguous()
        for _ in range(self.num_layers):
            out = ops.spmm_raw(graph.fwd, graph.w, None, out, y=res, a=self.alpha, b=1.0, kind="cs_propagate")
            out = post_step(out)
        return out


class CorrectAndSmooth:
    """Same constructor arguments as the PyG class (reference initial_params.py:42)."""

    def __init__(self, num_correction_layers, correction_alpha, num_smoothing_layers, smoothing_alpha,
                 autoscale=True, scale=1.0):
        self.autoscale, self.scale = autoscale, scale
        self.prop1 = LabelPropagation(num_correction_layers, correction_alpha)
        self.prop2 = LabelPropagation(num_smoothing_layers, smoothing_alpha)

    @staticmethod
    def _onehot(y_true, like):
        if y_true.dtype == torch.long:
            return F.one_hot(y_true.view(-1), like.size(-1)).to(like.dtype)
        return y_true

    @torch.no_grad()
    def correct(self, y_soft, y_true, mask, edge_index):
        """Spread the training residual: error = onehot - y_soft on `mask`, 0 elsewhere; propagate it
        (clamped to [-1, 1]); with autoscale, scale every row so its L1 norm equals the mean training
        residual norm (rows that would scale by inf or > 1000 keep scale 1)."""
        assert abs(float(y_soft.sum()) / y_soft.size(0) - 1.0) < 1e-2, "y_soft rows must be probabilities"
        numel = int(mask.sum()) if mask.dtype == torch.bool else mask.size(0)
        y_true = self._onehot(y_true, y_soft)
        error = torch.zeros_like(y_soft)
        error[mask] = y_true - y_soft[mask]
        if self.autoscale:
            smoothed = self.prop1(error, edge_index, post_step=lambda t: t.clamp_(-1.0, 1.0))
            sigma = error[mask].abs().sum() / numel
            scale = sigma / smoothed.abs().sum(dim=1, keepdim=True)
            scale[scale.isinf() | (scale > 1000)] = 1.0
            return y_soft + scale * smoothed

        def fix_input(t):
            t[mask] = error[mask]
            return t

        return y_soft + self.scale * self.prop1(error, edge_index, post_step=fix_input)

    @torch.no_grad()
    def smooth(self, y_soft, y_true, mask, edge_index):
        """Clamp the training rows to their labels and propagate (clamped to [0, 1])."""
        y_soft = y_soft.clone()
        y_soft[mask] = self._onehot(y_true, y_soft)
        return self.prop2(y_soft, edge_index)
