"""BatchNorm1d over the node axis on the HIP kernels (reference: nn.BatchNorm1d between conv layers,
models/gcn.py:23,28; graphsage.py:24,29; gat.py:23,29; appnp_stack.py:21,27). Same parameters, buffers
and state_dict keys as nn.BatchNorm1d. The column statistics are raw fp64 sums, which a node-partitioned
run all-reduces before finishing (dist.DistBatchNorm1d overrides `_reduce`)."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib


def _scratch(n, d, device):
    cnt = ctypes.c_int64(0)
    _lib.check(_lib.load().rgbx_bn_scratch_doubles(n, d, ctypes.byref(cnt)), "rgbx_bn_scratch_doubles")
    return torch.empty(cnt.value, dtype=torch.float64, device=device), cnt.value


def column_sums(x):
    """[2, d] float64: column sums of x and x^2."""
    if not x.is_cuda:  # host tensors (MLP on the CPU, gloo tests): plain torch, still fp64 accumulation
        xd = x.double()
        return torch.stack([xd.sum(0), (xd * xd).sum(0)])
    px, ldx = _lib.mat(x, "x")
    n, d = x.shape
    sums = torch.empty((2, d), dtype=torch.float64, device=x.device)
    scratch, cnt = _scratch(n, d, x.device)
    _lib.check(_lib.load().rgbx_bn_stats_f32(px, ldx, n, d, _lib.ptr(sums), _lib.ptr(scratch), cnt, _lib.stream_ptr()),
               "rgbx_bn_stats_f32")
    return sums


def affine_cols(x, scale, shift):
    if not x.is_cuda:
        return x * scale + shift
    px, ldx = _lib.mat(x, "x")
    y = torch.empty((x.size(0), x.size(1)), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rgbx_affine_cols_f32(px, ldx, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), y.stride(0),
                                                x.size(0), x.size(1), _lib.stream_ptr()), "rgbx_affine_cols_f32")
    return y


def bwd_sums(gy, x, mean, rstd):
    """[2, d] float64: column sums of gy and gy * xhat."""
    if not gy.is_cuda:
        xhat = (x - mean) * rstd
        return torch.stack([gy.double().sum(0), (gy * xhat).double().sum(0)])
    pg, ldg = _lib.mat(gy, "gy")
    px, ldx = _lib.mat(x, "x")
    n, d = x.shape
    sums = torch.empty((2, d), dtype=torch.float64, device=x.device)
    scratch, cnt = _scratch(n, d, x.device)
    _lib.check(_lib.load().rgbx_bn_bwd_reduce_f32(pg, ldg, px, ldx, _lib.ptr(mean), _lib.ptr(rstd), n, d, _lib.ptr(sums),
                                                  _lib.ptr(scratch), cnt, _lib.stream_ptr()), "rgbx_bn_bwd_reduce_f32")
    return sums


def bwd_apply(gy, x, mean, rstd, ca, cb, ck):
    if not gy.is_cuda:
        return (gy - ca - (x - mean) * rstd * cb) * ck
    pg, ldg = _lib.mat(gy, "gy")
    px, ldx = _lib.mat(x, "x")
    gx = torch.empty((x.size(0), x.size(1)), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rgbx_bn_bwd_apply_f32(pg, ldg, px, ldx, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ca),
                                                 _lib.ptr(cb), _lib.ptr(ck), _lib.ptr(gx), gx.stride(0), x.size(0),
                                                 x.size(1), _lib.stream_ptr()), "rgbx_bn_bwd_apply_f32")
    return gx


class _BNTrain(torch.autograd.Function):
    """Training-mode BatchNorm over the node axis. On the GPU: column sums (2 launches) -> [reduce over ranks] ->
    rgbx_bn_finalize_f32 (mean, rstd, affine map, running statistics: 1 launch) -> affine apply; backward: column
    sums -> [reduce] -> rgbx_bn_bwd_finalize_f32 -> apply. `running` = (running_mean, running_var, momentum) or
    None; on host tensors (MLP on the CPU, gloo tests) the same arithmetic in plain torch."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, reduce, running, colsums=None):
        x = x if x.stride(-1) == 1 else x.contiguous()
        mean, rstd, scale, shift, n = train_statistics(x, weight, bias, eps, reduce, running, colsums)
        y = affine_cols(x, scale.contiguous(), shift.contiguous())
        ctx.save_for_backward(x, weight, mean, rstd, n)
        ctx.reduce = reduce
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd, n = ctx.saved_tensors
        gx, gw, gb = train_backward(gy, x, weight, mean, rstd, n, ctx.reduce)
        return gx, gw, gb, None, None, None, None


def pack_statistics(x, colsums=None):
    """[sum x, sum x^2, rows] of this rank's rows as ONE float64 vector [2 d + 1]: what the ranks all-reduce."""
    d = x.size(1)
    # the row count rides along as a device scalar made by a fill kernel (a host->device copy would break
    # hipGraph capture)
    count = torch.full((1,), float(x.size(0)), dtype=torch.float64, device=x.device)
    sums = colsums if colsums is not None and tuple(colsums.shape) == (2, d) else column_sums(x)
    return torch.cat([sums.reshape(-1), count])


def train_statistics(x, weight, bias, eps, reduce, running, colsums=None, packed=None):
    """Training-mode statistics of x [N, d] (row-contiguous): (mean, rstd, scale, shift, n) with
    BN(x) = x * scale + shift, running statistics updated in place (`running` = (running_mean, running_var,
    momentum) or None). Column sums -> [reduce over ranks] -> one finalize launch. `colsums` ([2, d] float64): the
    column sums of x and x^2 when the kernel that produced x already took them (rgbx_spmm_linear_f32 out_colsums):
    the pass over x is then skipped. `packed`: pack_statistics' vector ALREADY reduced over the ranks (a caller that
    issues the all-reduce itself, asynchronously: dist/stack.py); `reduce` is then not called."""
    d = x.size(1)
    if packed is None:
        packed = reduce(pack_statistics(x, colsums))  # identity on one GPU; all-reduce over the node partition otherwise
    n = packed[2 * d:2 * d + 1]
    if x.is_cuda:
        mean, rstd, scale, shift = (torch.empty(d, dtype=torch.float32, device=x.device) for _ in range(4))
        rm, rv, mom = running if running is not None else (None, None, 0.0)
        _lib.check(_lib.load().rgbx_bn_finalize_f32(
            _lib.ptr(packed), _lib.ptr(weight.detach().contiguous()), _lib.ptr(bias.detach().contiguous()),
            float(eps), float(mom), _lib.ptr(rm), _lib.ptr(rv), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(scale),
            _lib.ptr(shift), d, _lib.stream_ptr()), "rgbx_bn_finalize_f32")
    else:
        mean64 = packed[:d] / n
        var64 = (packed[d:2 * d] / n - mean64 * mean64).clamp_(min=0.0)  # biased variance
        mean, var = mean64.float(), var64.float()
        rstd = torch.rsqrt(var + eps)
        scale = weight.detach() * rstd
        shift = bias.detach() - mean * scale
        if running is not None:
            rm, rv, mom = running
            unbiased = var * (n / (n - 1).clamp(min=1)).float()
            rm.mul_(1 - mom).add_(mean, alpha=mom)
            rv.mul_(1 - mom).add_(unbiased, alpha=mom)
    return mean, rstd, scale, shift, n


def train_backward(gy, x, weight, mean, rstd, n, reduce, grads_out=None, sums=None):
    """(gx, g_weight, g_bias) of training-mode BatchNorm given the gradient gy of its output. `grads_out` =
    (g_weight, g_bias) tensors to write the parameter gradients into (device path). `sums` = (local, global):
    bwd_sums of this rank's rows and their all-reduced copy when the caller made both (asynchronous all-reduce:
    dist/stack.py); `reduce` is then not called."""
    gy = gy if gy.stride(-1) == 1 else gy.contiguous()
    d = x.size(1)
    if sums is None:
        local = bwd_sums(gy, x, mean, rstd)            # [2, d]: sum gy, sum gy * xhat over the local rows
        glob = reduce(local.reshape(-1).clone())
    else:
        local, glob = sums
    # parameter gradients are the LOCAL sums: a partitioned run all-reduces parameter gradients once
    if gy.is_cuda:
        ca, cb, ck = (torch.empty(d, dtype=torch.float32, device=gy.device) for _ in range(3))
        gw, gb = grads_out if grads_out is not None else (torch.empty(d, dtype=torch.float32, device=gy.device)
                                                          for _ in range(2))
        _lib.check(_lib.load().rgbx_bn_bwd_finalize_f32(
            _lib.ptr(glob), _lib.ptr(local), _lib.ptr(n), _lib.ptr(weight.detach().contiguous()), _lib.ptr(rstd),
            _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(ck), _lib.ptr(gw), _lib.ptr(gb), d, _lib.stream_ptr()),
            "rgbx_bn_bwd_finalize_f32")
    else:
        glob = glob.reshape(2, -1)
        ca = (glob[0] / n).float().contiguous()
        cb = (glob[1] / n).float().contiguous()
        ck = (weight * rstd).contiguous()
        gw, gb = local[1].float(), local[0].float()
        if grads_out is not None:
            grads_out[0].copy_(gw)
            grads_out[1].copy_(gb)
            gw, gb = grads_out
    return bwd_apply(gy, x, mean, rstd, ca, cb, ck), gw, gb


class _AffineCols(torch.autograd.Function):
    """y = x * scale + shift per column (rgbx_affine_cols_f32) WITH a backward: gx = gy * scale,
    g_scale = sum_r gy * x, g_shift = sum_r gy. `scale` / `shift` are functions of the BatchNorm parameters
    (eval_affine), so weight and bias receive their gradients through them."""

    @staticmethod
    def forward(ctx, x, scale, shift):
        x = x if x.stride(-1) == 1 else x.contiguous()
        ctx.save_for_backward(x, scale)
        return affine_cols(x, scale.detach().contiguous(), shift.detach().contiguous())

    @staticmethod
    def backward(ctx, gy):
        x, scale = ctx.saved_tensors
        gy = gy if gy.stride(-1) == 1 else gy.contiguous()
        gx = g_scale = g_shift = None
        if ctx.needs_input_grad[0]:
            gx = affine_cols(gy, scale.detach().contiguous(), torch.zeros_like(scale))
        if ctx.needs_input_grad[1]:
            g_scale = (gy * x).sum(0)
        if ctx.needs_input_grad[2]:
            g_shift = gy.sum(0)
        return gx, g_scale, g_shift


class BatchNorm1d(nn.BatchNorm1d):
    """Drop-in for nn.BatchNorm1d on [N, d] inputs (affine, running statistics)."""

    def _reduce(self, packed):
        return packed

    def _reducer(self, x):
        """The reduction of the raw column sums that goes with THIS input (dist.DistBatchNorm1d: all-reduce over the
        node partition, or none when every rank holds all rows of x)."""
        return self._reduce

    def eval_affine(self):
        """(scale, shift) with eval-mode BN(x) = x * scale + shift, or None when that form does not apply."""
        if self.training or not self.affine or not self.track_running_stats:
            return None
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        return scale, self.bias - self.running_mean * scale

    def forward(self, x, colsums=None):
        """`colsums`: see train_statistics (training forwards only)."""
        if x.dim() != 2 or not self.affine or not self.track_running_stats:
            return super().forward(x)
        if not self.training:
            scale, shift = self.eval_affine()
            if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or self.bias.requires_grad):
                # eval mode with autograd on (frozen-BN fine-tuning, saliency on the inputs): nn.BatchNorm1d is
                # differentiable there, so this must be too — the raw kernel below has no autograd node
                return _AffineCols.apply(x, scale, shift)
            return affine_cols(x if x.stride(-1) == 1 else x.contiguous(), scale.contiguous(), shift.contiguous())
        return _BNTrain.apply(x, self.weight, self.bias, self.eps, self._reducer(x), self.begin_training_step(), colsums)

    def begin_training_step(self):
        """What nn.BatchNorm1d does at the top of a training forward: count the batch, pick the momentum.
        Returns the `running` triple (running_mean, running_var, momentum) for train_statistics."""
        with torch.no_grad():
            self.num_batches_tracked += 1
        if self.momentum is None:  # cumulative moving average: the factor depends on a device counter
            m = 1.0 / float(self.num_batches_tracked)
        else:
            m = self.momentum
        return self.running_mean, self.running_var, m

    def folds_into_next_layer(self, x):
        """Training forward on the GPU with the standard affine / running-statistics set-up: the normalised
        matrix need not be written, the next conv layer can apply the affine map to its aggregate
        (ops.bn_propagate_linear)."""
        return (self.training and self.affine and self.track_running_stats and x.dim() == 2 and x.is_cuda
                and torch.is_grad_enabled())
