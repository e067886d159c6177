"""BatchNorm1d over the node axis with statistics reduced across the node partition, so that logits
equal the single-GPU ones (reference models/gcn.py:23,28 normalises over ALL N nodes)."""
import torch
import torch.nn as nn


class _DistBNTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, comm, stats_out):
        d = x.size(1)
        buf = torch.zeros(d + 1, dtype=torch.float32, device=x.device)
        buf[:d] = x.sum(0)
        buf[d] = x.size(0)
        comm.all_reduce_sum_(buf)
        n = buf[d]
        mean = buf[:d] / n
        xc = x - mean
        var = comm.all_reduce_sum_((xc * xc).sum(0)) / n  # biased, two-pass
        rstd = torch.rsqrt(var + eps)
        xhat = xc * rstd
        stats_out.extend([mean, var, n])
        ctx.save_for_backward(xhat, weight, rstd, n)
        ctx.comm = comm
        return xhat * weight + bias

    @staticmethod
    def backward(ctx, gy):
        xhat, weight, rstd, n = ctx.saved_tensors
        d = gy.size(1)
        local = torch.cat([gy.sum(0), (gy * xhat).sum(0)])
        glob = ctx.comm.all_reduce_sum_(local.clone())
        gx = (weight * rstd) * (gy - glob[:d] / n - xhat * (glob[d:] / n))
        # parameter gradients stay LOCAL sums: the runner all-reduces every parameter gradient once
        return gx, local[d:], local[:d], None, None, None


class DistBatchNorm1d(nn.BatchNorm1d):
    """Same parameters / buffers / state_dict keys as nn.BatchNorm1d."""

    def __init__(self, num_features, comm, **kw):
        super().__init__(num_features, **kw)
        self.comm = comm

    @classmethod
    def convert(cls, module, comm):
        """Replace every nn.BatchNorm1d under `module` (in place), keeping parameters and buffers."""
        for name, child in list(module.named_children()):
            if isinstance(child, nn.BatchNorm1d) and not isinstance(child, cls):
                new = cls(child.num_features, comm, eps=child.eps, momentum=child.momentum)
                new.load_state_dict(child.state_dict())
                new.to(child.weight.device)
                new.train(child.training)
                setattr(module, name, new)
            else:
                cls.convert(child, comm)
        return module

    def forward(self, x):
        if not self.training:
            return super().forward(x)
        stats = []
        y = _DistBNTrain.apply(x, self.weight, self.bias, self.eps, self.comm, stats)
        mean, var, n = stats
        with torch.no_grad():
            m = self.momentum
            self.num_batches_tracked += 1
            self.running_mean.mul_(1 - m).add_(mean, alpha=m)
            self.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp(min=1)), alpha=m)
        return y
