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
