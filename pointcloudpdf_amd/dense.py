"""Dense per-point layers (nn.Linear on (N, C) with N ~ 10^5, C <= 512).

Forward and input-gradient GEMMs are plain library GEMMs (rocBLAS/hipBLASLt through torch: 23 us for 200k x 32 x 32).
The WEIGHT gradient dW = dY^T X is a (C_out x N) x (N x C_in) product with a tiny output: the library runs it as a
single 32x32 macro-tile with K = N on ONE workgroup (measured 270 us per call, 44 calls per step).  ``linear`` below
keeps nn.Linear's parameters but routes that one product through a batched GEMM over row chunks (split-K expressed
as a batch: S = N / 2048 independent products, summed afterwards), which fills the chip.
"""
import torch
import torch.nn.functional as F

_CHUNK = 2048
_MIN_ROWS = 16384


class _LinearSplitK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            n, ci = x.shape
            co = g.shape[1]
            s = n // _CHUNK
            body = s * _CHUNK
            gw = torch.bmm(g[:body].view(s, _CHUNK, co).transpose(1, 2), x[:body].view(s, _CHUNK, ci)).sum(0)
            if body < n:
                gw = gw + g[body:].t() @ x[body:]
        gb = g.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def linear(module, x):
    """``module(x)`` for an nn.Linear, with the split-K weight gradient when the row count is large."""
    if x.dim() == 2 and x.is_cuda and x.shape[0] >= _MIN_ROWS and torch.is_grad_enabled() and x.is_contiguous():
        return _LinearSplitK.apply(x, module.weight, module.bias)
    return module(x)


class _BnAct(torch.autograd.Function):
    """relu?( BatchNorm1d(x) [+ residual] ) as one autograd node over the HIP kernels of csrc/pointwise.hip."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, bn, relu):
        from . import _native

        be = _native.hip_backend()
        training = bn.training or bn.running_mean is None
        y, coef = be.bn_act_forward(x, residual, weight.detach(), bias.detach(), bn.running_mean, bn.running_var, training,
                                    bn.eps, bn.momentum if bn.momentum is not None else 0.1, relu)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x, residual if residual is not None else x.new_empty(0), coef)
        ctx.cfg = (training, relu, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _native

        x, residual, coef = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        be = _native.hip_backend()
        gx, gres, ggamma, gbeta = be.bn_act_backward(gy.contiguous(), x, residual if has_res else None, coef, training, relu,
                                                     has_res and ctx.needs_input_grad[1])
        return gx, gres, ggamma, gbeta, None, None


def bn_act(bn, x, residual=None, relu=True):
    """``relu(bn(x) + residual)`` for an nn.BatchNorm1d on (N, C) rows; falls back to torch off-device / odd widths."""
    from . import _native

    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and bn.affine
            and bn.track_running_stats and _native.hip_backend().bn_supported(x.shape[1])
            and (residual is None or residual.is_contiguous())):
        return _BnAct.apply(x, residual, bn.weight, bn.bias, bn, relu)
    y = torch.nn.BatchNorm1d.forward(bn, x)  # (not bn(x): subclasses such as LayerNorm1d route back here)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y
