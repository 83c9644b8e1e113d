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


# ------------------------------------------------------------------------------------------------------------------
# BatchNorm step counters.  nn.BatchNorm increments ``num_batches_tracked`` once per training forward: one tiny kernel per
# norm, ~80 launches per step on this model.  Inside ``deferred_counters()`` (the training step wraps its forward in it) the
# increments are collected and applied with ONE multi-tensor add on exit; outside, every norm bumps its counter at once.
# ------------------------------------------------------------------------------------------------------------------
_pending_counters = None


def bump_counters(counters):
    counters = [c for c in counters if c is not None]
    if not counters:
        return
    if _pending_counters is not None:
        _pending_counters.extend(counters)
    elif len(counters) == 1:
        counters[0].add_(1)
    else:
        torch._foreach_add_(counters, 1)


class deferred_counters:
    def __enter__(self):
        global _pending_counters
        self._outer = _pending_counters
        _pending_counters = []
        return self

    def __exit__(self, *exc):
        global _pending_counters
        mine, _pending_counters = _pending_counters, self._outer
        if mine:
            if self._outer is not None:
                self._outer.extend(mine)
            else:
                by_dev = {}
                for c in mine:
                    by_dev.setdefault(c.device, []).append(c)
                for cs in by_dev.values():
                    torch._foreach_add_(cs, 1)
        return False


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
        if training:
            bump_counters([bn.num_batches_tracked])
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


# ------------------------------------------------------------------------------------------------------------------
# Matrix-core Linear layers with the neighbouring BatchNorm folded in (csrc/rowlin.hip).  One autograd node each:
#   _LinearStats   : y = x W^T + b                     (+ column statistics of y for the BatchNorm that follows)
#   _BnReluLinear  : y_i = relu(bn(z)) W_i^T + b_i      (statistics of z from the producer's epilogue or a stats pass;
#                                                         the normalised activation never reaches HBM)
#   _BnActPartial  : relu(bn(z) + residual) with the statistics of z supplied by the producer's epilogue
# ------------------------------------------------------------------------------------------------------------------
def _be():
    from . import _native

    return _native.hip_backend()


class _LinearStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, want_stats):
        y, partial = _be().rowlin(x, weight.detach(), None if bias is None else bias.detach(), stats=want_stats)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if partial is None:
            partial = x.new_empty(0)
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, g, _gp):
        x, weight = ctx.saved_tensors
        be = _be()
        g = g.contiguous()
        gx = be.rowlin(g, weight, transpose_w=True)[0] if ctx.needs_input_grad[0] else None
        gw, gb = be.rowlin_wgrad(g, x, None, False, ctx.has_bias)
        return gx, gw, gb, None


class _BnReluLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, partial, gamma, beta, bn, relu, want_stats, *wb):
        be = _be()
        n, c = z.shape
        training = bn.training
        if training and partial is not None and partial.numel() > 0:
            coef = be.bn_coef_from_partial(partial, n, c, bn, True)
        else:
            coef = be.bn_coef(z, bn, training)
        if training:
            bump_counters([bn.num_batches_tracked])
        weights, biases = wb[0::2], wb[1::2]
        outs, pout = [], None
        for w, b in zip(weights, biases):
            y, p = be.rowlin(z, w.detach(), None if b is None else b.detach(), coef=coef, relu=relu, stats=want_stats)
            outs.append(y)
            pout = p
        ctx.save_for_backward(z, coef, *weights)
        ctx.cfg = (training, relu, [b is not None for b in biases])
        if pout is None:
            pout = z.new_empty(0)
        ctx.mark_non_differentiable(pout)
        return (*outs, pout)

    @staticmethod
    def backward(ctx, *gs):
        z, coef, *weights = ctx.saved_tensors
        training, relu, has_bias = ctx.cfg
        be = _be()
        grads_wb = []
        da = None
        for i, (g, w) in enumerate(zip(gs[:-1], weights)):
            if g is None:
                grads_wb += [torch.zeros_like(w), torch.zeros(w.shape[0], device=w.device) if has_bias[i] else None]
                continue
            g = g.contiguous()
            da = be.rowlin(g, w, transpose_w=True, out=da, accumulate=da is not None)[0]
            gw, gb = be.rowlin_wgrad(g, z, coef, relu, has_bias[i])
            grads_wb += [gw, gb]
        gz, _, ggamma, gbeta = be.bn_act_backward(da, z, None, coef, training, relu, False)
        return (gz, None, ggamma, gbeta, None, None, None, *grads_wb)


class _BnActPartial(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, partial, residual, gamma, beta, bn, relu):
        be = _be()
        n, c = z.shape
        training = bn.training
        if training and partial is not None and partial.numel() > 0:
            coef = be.bn_coef_from_partial(partial, n, c, bn, True)
        else:
            coef = be.bn_coef(z, bn, training)
        if training:
            bump_counters([bn.num_batches_tracked])
        y = be.bn_apply(z, residual, coef, relu)
        ctx.save_for_backward(z, residual if residual is not None else z.new_empty(0), coef)
        ctx.cfg = (training, relu, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        z, residual, coef = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        gz, gres, ggamma, gbeta = _be().bn_act_backward(gy.contiguous(), z, residual if has_res else None, coef, training, relu,
                                                       has_res and ctx.needs_input_grad[2])
        return gz, None, gres, ggamma, gbeta, None, None


def fused_ok(x, *bns):
    """The matrix-core path applies to fp32 (N, C) rows on the device with power-of-two BatchNorm widths."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()):
        return False
    be = _be()
    return all(type(b) is torch.nn.BatchNorm1d and b.affine and b.track_running_stats and be.bn_supported(b.num_features) for b in bns)


def linear_stats(lin, x, want_stats=True):
    y, partial = _LinearStats.apply(x, lin.weight, lin.bias, want_stats)
    return y, partial


def bn_relu_linear(bn, z, partial, lins, relu=True, want_stats=False):
    wb = []
    for lin in lins:
        wb += [lin.weight, lin.bias]
    out = _BnReluLinear.apply(z, partial, bn.weight, bn.bias, bn, relu, want_stats, *wb)
    return out[:-1], out[-1]


def bn_act_partial(bn, z, partial, residual=None, relu=True):
    return _BnActPartial.apply(z, partial, residual, bn.weight, bn.bias, bn, relu)


# ------------------------------------------------------------------------------------------------------------------
# Bottleneck halves: ONE host call per half and direction (csrc/block.hip).
# ------------------------------------------------------------------------------------------------------------------
class _BlockPre(torch.autograd.Function):
    """(x_q, x_k, x_v) = q/k/v( relu(bn1(linear1(x))) )"""

    @staticmethod
    def forward(ctx, x, W1, g1, b1, Wq, bq, Wk, bk, Wv, bv, blk):
        be = _be()
        n, c = x.shape
        bn1, training = blk.bn1, blk.bn1.training
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        z1, coef1, xq, xk, xv = e(n, c), e(4 * c), e(n, c), e(n, c), e(n, c)
        partial = e(int(be.lib.pdf_rowlin_partial_floats(n, c)))
        be.block_call("pre_forward", n, c, [x, W1, g1, b1, bn1.running_mean, bn1.running_var, Wq, bq, Wk, bk, Wv, bv,
                                            z1, coef1, xq, xk, xv, partial], training, bn1.eps, bn1.momentum or 0.1)
        ctx.save_for_backward(x, z1, coef1, W1, Wq, Wk, Wv)
        ctx.training = training
        return xq, xk, xv

    @staticmethod
    def backward(ctx, gxq, gxk, gxv):
        x, z1, coef1, W1, Wq, Wk, Wv = ctx.saved_tensors
        be = _be()
        n, c = x.shape
        cc = c * c
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        gx, grads, dy, sums = e(n, c), e(cc + 2 * c + 3 * (cc + c)), e(n, c), e(2 * c)
        partial = e(int(be.lib.pdf_bn_partial_floats(n, c)))
        be.block_call("pre_backward", n, c, [x, z1, coef1, W1, Wq, Wk, Wv, gxq.contiguous(), gxk.contiguous(), gxv.contiguous(),
                                             gx, grads, dy, partial, sums], ctx.training)
        o = cc + 2 * c
        out = [gx, grads[:cc].view(c, c), grads[cc + c:cc + 2 * c], grads[cc:cc + c]]   # dW1, dgamma1, dbeta1 (buffer: dW1 | dbeta1 | dgamma1)
        for i in range(3):
            out += [grads[o + i * (cc + c): o + i * (cc + c) + cc].view(c, c), grads[o + i * (cc + c) + cc: o + (i + 1) * (cc + c)]]
        return (*out, None)


class _BlockPost(torch.autograd.Function):
    """y = relu( bn3(linear3(relu(bn2(t)))) + x )"""

    @staticmethod
    def forward(ctx, t, x, g2, b2, W3, g3, b3, blk):
        be = _be()
        n, c = t.shape
        bn2, bn3, training = blk.bn2, blk.bn3, blk.bn2.training
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=t.device)
        coef2, z3, coef3, y = e(4 * c), e(n, c), e(4 * c), e(n, c)
        partial = e(max(int(be.lib.pdf_rowlin_partial_floats(n, c)), int(be.lib.pdf_bn_partial_floats(n, c))))
        be.block_call("post_forward", n, c, [t, x, g2, b2, bn2.running_mean, bn2.running_var, W3, g3, b3, bn3.running_mean,
                                             bn3.running_var, coef2, z3, coef3, y, partial], training, bn2.eps, bn2.momentum or 0.1)
        ctx.save_for_backward(t, x, z3, coef2, coef3, W3)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, gy):
        t, x, z3, coef2, coef3, W3 = ctx.saved_tensors
        be = _be()
        n, c = t.shape
        cc = c * c
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=t.device)
        gt, gres, grads, da, sums = e(n, c), e(n, c), e(cc + 4 * c), e(n, c), e(2 * c)
        partial = e(int(be.lib.pdf_bn_partial_floats(n, c)))
        be.block_call("post_backward", n, c, [gy.contiguous(), t, x, z3, coef2, coef3, W3, gt, gres, grads, da, partial, sums],
                      ctx.training)
        # buffer: dW3 | dbeta2 | dgamma2 | dbeta3 | dgamma3 ; forward args: t, x, g2, b2, W3, g3, b3
        return (gt, gres, grads[cc + c:cc + 2 * c], grads[cc:cc + c], grads[:cc].view(c, c), grads[cc + 3 * c:cc + 4 * c],
                grads[cc + 2 * c:cc + 3 * c], None)


def bottleneck(blk, p, x, o):
    """Bottleneck.forward through the two host-side halves and the (fused) attention layer."""
    t = blk.transformer
    xq, xk, xv = _BlockPre.apply(x, blk.linear1.weight, blk.bn1.weight, blk.bn1.bias, t.linear_q.weight, t.linear_q.bias,
                                 t.linear_k.weight, t.linear_k.bias, t.linear_v.weight, t.linear_v.bias, blk)
    a = t.attend(p, x, o, xq, xk, xv)
    y = _BlockPost.apply(a.contiguous(), x, blk.bn2.weight, blk.bn2.bias, blk.linear3.weight, blk.bn3.weight, blk.bn3.bias, blk)
    if blk.training:
        bump_counters([blk.bn1.num_batches_tracked, blk.bn2.num_batches_tracked, blk.bn3.num_batches_tracked])
    return y
