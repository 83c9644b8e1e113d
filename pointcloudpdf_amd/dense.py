"""Dense per-point layers (nn.Linear on (N, C) with N ~ 10^5, C <= 512).

Forward and input-gradient GEMMs are plain library GEMMs (rocBLAS/hipBLASLt through torch: 23 us for 200k x 32 x 32).
The WEIGHT gradient dW = dY^T X is a (C_out x N) x (N x C_in) product with a tiny output: the library runs it as a
single 32x32 macro-tile with K = N on ONE workgroup (measured 270 us per call, 44 calls per step).  ``linear`` below
keeps nn.Linear's parameters but routes that one product through a batched GEMM over row chunks (split-K expressed
as a batch: S = N / 2048 independent products, summed afterwards), which fills the chip.
"""
import ctypes
import os
from ctypes import c_void_p

import torch

# Custom autograd nodes keep fp32 tensors under autocast (upstream's python ops promote to fp32 the same way); a node remembers the
# product-input mode of its forward (fp32 / fp16 / bfloat16 operands of the streaming Linear kernels, _native.mma_input) and restores it
# for its backward, which runs outside the autocast region.
def _amp_fwd(fn):
    import functools

    from . import _native
    inner = torch.amp.custom_fwd(fn, device_type="cuda", cast_inputs=torch.float32)

    @functools.wraps(fn)
    def wrapped(ctx, *args, **kwargs):
        ctx._pdf_mma = _native.current_mma_input()
        return inner(ctx, *args, **kwargs)

    return wrapped


def _amp_bwd(fn):
    import functools

    from . import _native
    inner = torch.amp.custom_bwd(fn, device_type="cuda")

    @functools.wraps(fn)
    def wrapped(ctx, *grads):
        with _native.mma_input(getattr(ctx, "_pdf_mma", 0)):
            return inner(ctx, *grads)

    return wrapped


import torch.nn as nn
import torch.nn.functional as F


def _mma():
    """product-input mode of the calling thread (the `mma_input` argument of the C entry points, _native.mma_input)"""
    from . import _native
    return _native.current_mma_input()


_CHUNK = 2048
_MIN_ROWS = 16384


amp_mma = os.environ.get("PDFOPS_AMP_MMA", "1") != "0"   # 0: autocast leaves the products in fp32 as well (bit-identical to no autocast)


def fp32_path(fn):
    """Decorator of the model-level forwards.  Every tensor of the path is fp32 (BatchNorm statistics, gathers, attention and all
    accumulators included), so under ``torch.autocast`` -- the reference's trainer runs with ``enable_amp = True``
    (configs/s3dis/openseg-pt-v1-0-msp.py:6, engines/train.py:340-363) -- the modules opt OUT of autocast's per-op casts (measured in
    round 2: wrapping the remaining torch ops in half precision and the custom nodes in casts costs 30.9 ms vs 17 ms per step) and run
    the reduced-precision variant of the path instead: the operands of the streaming Linear products (forward, input gradient, weight
    gradient) are rounded to the autocast dtype (fp16 / bfloat16) in registers and multiplied on the 16x16x16 matrix-core instructions,
    with fp32 accumulation (csrc/rowlin2_impl.h: Mma; ``_native.mma_input``).  Geometry (kNN / FPS tables) is unaffected; logits move by
    the rounding of the operands (~1e-3).  ``PDFOPS_AMP_MMA=0`` / ``dense.amp_mma = False`` keeps fp32 operands: autocast then changes
    nothing at all."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if torch.is_autocast_enabled("cuda"):
            from . import _native

            mode = _native.MMA_INPUT_OF_DTYPE.get(torch.get_autocast_dtype("cuda"), 0) if amp_mma else 0
            with torch.autocast("cuda", enabled=False), _native.mma_input(mode):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)

    return wrapped


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


_ONES_ROWS = {}


def _ones_row(device, dtype):
    key = (device, dtype)
    if key not in _ONES_ROWS:
        _ONES_ROWS[key] = torch.ones(1, 1, _CHUNK, device=device, dtype=dtype)
    return _ONES_ROWS[key]


class _LinearSplitK(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        if (_OWN_WGRAD and ctx.needs_input_grad[1] and g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and x.dim() == 2
                and x.stride(1) == 1):
            # weight AND bias gradient as one split-K launch + one reduction of the package's own matrix-core kernel (csrc/rowlin*.hip):
            # 26-90 us per call at the StratifiedTransformer's shapes against 96-108 us of the five-launch library composition below
            # (tools/probes/st_linear_probe.py; forward and input gradient stay on the library GEMMs, which win there)
            from . import _native

            gw, gb = _native.backend_for(g).rowlin_wgrad(g, x, None, False, ctx.has_bias and ctx.needs_input_grad[2])
            return gx, gw, gb
        gw = None
        if ctx.needs_input_grad[1]:
            n, ci = x.shape
            co = g.shape[1]
            s = n // _CHUNK
            body = s * _CHUNK
            gw = torch.bmm(g[:body].view(s, _CHUNK, co).transpose(1, 2), x[:body].view(s, _CHUNK, ci)).sum(0)
            if body < n:
                gw = gw + g[body:].t() @ x[body:]
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # the bias gradient as a batched product with a row of ones (same split over the rows): torch's column reduction of a
            # (200000, 13) tensor takes 250 us on MI355X, this 35
            n, co = g.shape
            s = n // _CHUNK
            body = s * _CHUNK
            gb = torch.bmm(_ones_row(g.device, g.dtype).expand(s, 1, _CHUNK), g[:body].view(s, _CHUNK, co)).sum(0).view(co) if s > 0 else g.new_zeros(co)
            if body < n:
                gb = gb + g[body:].sum(0)
        return gx, gw, gb


_OWN_WGRAD = os.environ.get("PDFOPS_SPLITK_WGRAD", "own") != "torch"   # torch: the library composition (A/B runs)
HIP_LINEAR = os.environ.get("PDFOPS_LINEAR", "hip") != "torch"   # torch: library GEMMs + split-K weight gradient (rounds 1-2; A/B runs)


def linear(module, x):
    """``module(x)`` for an nn.Linear outside the fused Linear -> BatchNorm nodes (first layer 6 -> 32, the 32 -> 13 / 32 -> 1 heads, the
    1024 -> 512 TransitionUp head): forward, input gradient and weight gradient through the library's own matrix-core kernels
    (csrc/rowlin.hip: the tiled kernel takes any (k, o); no library GEMM is left in the step).  ``PDFOPS_LINEAR=torch`` restores the
    library GEMMs with the split-K weight gradient (``_LinearSplitK``)."""
    if (HIP_LINEAR and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and module.weight.dtype == torch.float32
            and module.weight.is_contiguous()):
        return _LinearStats.apply(x, module.weight, module.bias, False)[0]
    if x.dim() == 2 and x.is_cuda and x.shape[0] >= _MIN_ROWS and torch.is_grad_enabled() and x.is_contiguous():
        return _LinearSplitK.apply(x, module.weight, module.bias)
    return module(x)


class _BnAct(torch.autograd.Function):
    """relu?( BatchNorm1d(x) [+ residual] ) as one autograd node over the HIP kernels of csrc/pointwise.hip."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, residual, weight, bias, bn, relu):
        from . import _native

        be = _native.hip_backend()
        training = bn.training or bn.running_mean is None
        y, coef = be.bn_act_forward(x, residual, weight.detach(), bias.detach(), bn.running_mean, bn.running_var, training,
                                    bn.eps, bn.momentum, relu)
        if training:
            bump_counters([bn.num_batches_tracked])
        ctx.save_for_backward(x, residual if residual is not None else x.new_empty(0), coef)
        ctx.cfg = (training, relu, residual is not None)
        return y

    @staticmethod
    @_amp_bwd
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
            and bn.track_running_stats and bn.momentum is not None and _native.hip_backend().bn_supported(x.shape[1])
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


def _rcd(*tensors):
    """All tensors of a fused node live on the current device (the launches go onto its current stream)."""
    from . import _native

    _native.require_current_device(*tensors)


class _LinearStats(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias, want_stats):
        _rcd(x, weight)
        y, partial = _be().rowlin(x, weight.detach(), None if bias is None else bias.detach(), stats=want_stats)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if partial is None:
            partial = x.new_empty(0)
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    @_amp_bwd
    def backward(ctx, g, _gp):
        x, weight = ctx.saved_tensors
        be = _be()
        g = g.contiguous()
        gx = be.rowlin(g, weight, transpose_w=True)[0] if ctx.needs_input_grad[0] else None
        gw, gb = be.rowlin_wgrad(g, x, None, False, ctx.has_bias)
        return gx, gw, gb, None


class _BnReluLinear(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, z, partial, gamma, beta, bn, relu, want_stats, *wb):
        be = _be()
        _rcd(z, gamma, wb[0])
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
    @_amp_bwd
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
    @_amp_fwd
    def forward(ctx, z, partial, residual, gamma, beta, bn, relu):
        be = _be()
        _rcd(z, residual, gamma)
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
    @_amp_bwd
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
    # (momentum=None means a cumulative moving average in torch: the fused norms implement the exponential update only)
    return all(type(b) is torch.nn.BatchNorm1d and b.affine and b.track_running_stats and b.momentum is not None
               and be.bn_supported(b.num_features) for b in bns)


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
# torch.nn.LayerNorm on (n, c) rows as one pass per direction (csrc/layernorm.hip) -- StratifiedTransformer's norms.
# ------------------------------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias, eps):
        be = _be()
        n, c = x.shape
        y = torch.empty_like(x)
        stat = torch.empty((2, n), dtype=torch.float32, device=x.device)
        rc = be.lib.pdf_layernorm_forward(n, c, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), ctypes.c_float(eps), y.data_ptr(),
                                          stat[0].data_ptr(), stat[1].data_ptr(), be._stream())
        if rc != 0:
            raise RuntimeError(f"pdf_layernorm_forward failed with status {rc}")
        ctx.save_for_backward(x, stat, weight)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        x, stat, weight = ctx.saved_tensors
        be = _be()
        n, c = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
        partial = torch.empty((max(int(be.lib.pdf_layernorm_partial_floats(n, c)), 1),), dtype=torch.float32, device=x.device)
        rc = be.lib.pdf_layernorm_backward(n, c, gy.data_ptr(), x.data_ptr(), stat[0].data_ptr(), stat[1].data_ptr(), weight.data_ptr(),
                                           gx.data_ptr(), partial.data_ptr(), gwb[0].data_ptr(), gwb[1].data_ptr(), be._stream())
        if rc != 0:
            raise RuntimeError(f"pdf_layernorm_backward failed with status {rc}")
        return gx, gwb[0], gwb[1], None


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters and ``state_dict`` keys) whose forward / backward on fp32 device rows run as ONE HIP pass each
    (csrc/layernorm.hip) instead of torch's three kernels; anything else (CPU, other dtypes, no affine parameters) is torch's."""

    hip = os.environ.get("PDFOPS_LAYERNORM", "hip") == "hip"

    def forward(self, x):
        c = x.shape[-1]
        if (self.hip and x.is_cuda and x.dtype == torch.float32 and self.weight is not None and self.bias is not None and len(self.normalized_shape) == 1
                and x.numel() > 0 and x.is_contiguous() and x.data_ptr() % 16 == 0 and _be().lib.pdf_layernorm_supported(c)):
            return _LayerNormFn.apply(x.view(-1, c), self.weight, self.bias, float(self.eps)).view(x.shape)
        return super().forward(x)


# ------------------------------------------------------------------------------------------------------------------
# Bottleneck halves: ONE host call per half and direction (csrc/block.hip).
# ------------------------------------------------------------------------------------------------------------------
class _BlockPre(torch.autograd.Function):
    """(x_q, x_k, x_v) = q/k/v( relu(bn1(linear1(x))) )"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, W1, g1, b1, Wq, bq, Wk, bk, Wv, bv, blk):
        be = _be()
        _rcd(x, W1)
        n, c = x.shape
        bn1, training = blk.bn1, blk.bn1.training
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        z1, coef1, xq, xk, xv = e(n, c), e(4 * c), e(n, c), e(n, c), e(n, c)
        partial = e(int(be.lib.pdf_rowlin_partial_floats(n, c)))
        be.block_call("pre_forward", n, c, [x, W1, g1, b1, bn1.running_mean, bn1.running_var, Wq, bq, Wk, bk, Wv, bv,
                                            z1, coef1, xq, xk, xv, partial], training, bn1.eps, bn1.momentum)
        ctx.save_for_backward(x, z1, coef1, W1, Wq, Wk, Wv)
        ctx.training = training
        return xq, xk, xv

    @staticmethod
    @_amp_bwd
    def backward(ctx, gxq, gxk, gxv):
        x, z1, coef1, W1, Wq, Wk, Wv = ctx.saved_tensors
        be = _be()
        n, c = x.shape
        cc = c * c
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        gx, grads, dy = e(n, c), e(cc + 2 * c + 3 * (cc + c)), e(n, c)
        # pdf_rowlin_dgrad_bstats writes pdf_rowlin_partial_rows rows of 2c floats here, the BatchNorm passes pdf_bn_partial_floats
        partial = e(max(int(be.lib.pdf_bn_partial_floats(n, c)), int(be.lib.pdf_rowlin_partial_floats(n, c))))
        ws = be.wgrad_workspace(n, c, c, 3, x.device)   # (also holds the slabs of the single-gradient dW1 product that follows)
        ws1 = int(be.lib.pdf_rowlin_wgrad_ws_floats(n, c, c, 1))
        if ws1 > ws.numel():
            ws = e(ws1)
        be.block_call("pre_backward", n, c, [x, z1, coef1, W1, Wq, Wk, Wv, gxq.contiguous(), gxk.contiguous(), gxv.contiguous(),
                                             gx, grads, dy, partial, ws], ctx.training)
        o = cc + 2 * c
        out = [gx, grads[:cc].view(c, c), grads[cc + c:cc + 2 * c], grads[cc:cc + c]]   # dW1, dgamma1, dbeta1 (buffer: dW1 | dbeta1 | dgamma1)
        for i in range(3):
            out += [grads[o + i * (cc + c): o + i * (cc + c) + cc].view(c, c), grads[o + i * (cc + c) + cc: o + (i + 1) * (cc + c)]]
        return (*out, None)


class _BlockPost(torch.autograd.Function):
    """y = relu( bn3(linear3(relu(bn2(t)))) + x )"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, t, x, g2, b2, W3, g3, b3, blk):
        be = _be()
        _rcd(t, x, W3)
        n, c = t.shape
        bn2, bn3, training = blk.bn2, blk.bn3, blk.bn2.training
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=t.device)
        coef2, z3, coef3, y = e(4 * c), e(n, c), e(4 * c), e(n, c)
        partial = e(max(int(be.lib.pdf_rowlin_partial_floats(n, c)), int(be.lib.pdf_bn_partial_floats(n, c))))
        be.block_call("post_forward", n, c, [t, x, g2, b2, bn2.running_mean, bn2.running_var, W3, g3, b3, bn3.running_mean,
                                             bn3.running_var, coef2, z3, coef3, y, partial], training, bn2.eps, bn2.momentum)
        ctx.save_for_backward(t, x, z3, coef2, coef3, W3)
        ctx.training = training
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        t, x, z3, coef2, coef3, W3 = ctx.saved_tensors
        be = _be()
        n, c = t.shape
        cc = c * c
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=t.device)
        gt, gres, grads, da = e(n, c), e(n, c), e(cc + 4 * c), e(n, c)
        partial = e(max(int(be.lib.pdf_bn_partial_floats(n, c)), int(be.lib.pdf_rowlin_partial_floats(n, c))))
        ws = be.wgrad_workspace(n, c, c, 1, t.device)
        be.block_call("post_backward", n, c, [gy.contiguous(), t, x, z3, coef2, coef3, W3, gt, gres, grads, da, partial, ws],
                      ctx.training)
        # buffer: dW3 | dbeta2 | dgamma2 | dbeta3 | dgamma3 ; forward args: t, x, g2, b2, W3, g3, b3
        return (gt, gres, grads[cc + c:cc + 2 * c], grads[cc:cc + c], grads[:cc].view(c, c), grads[cc + 3 * c:cc + 4 * c],
                grads[cc + 2 * c:cc + 3 * c], None)


# ------------------------------------------------------------------------------------------------------------------
# The whole Bottleneck as ONE autograd node and ONE host call per direction (csrc/block.hip: pdf_bottleneck_*): the forward
# of the step is host-bound, and three Function.apply + ~25 allocations + three ctypes calls per block were most of it.
# ------------------------------------------------------------------------------------------------------------------
def _al(v):
    return (v + 63) & ~63   # 256-byte aligned sub-buffers


class _BottleneckFn(torch.autograd.Function):
    """params: W1 g1 b1 | Wq bq Wk bk Wv bv | layer: Wp1 bp1 gp bp Wp2 bp2 g1' b1' Ww1 bw1 g2' b2' Ww2 bw2 | g2 b2 W3 g3 b3"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, blk, idx, p, x, *params):
        be = _be()
        _rcd(idx, p, x, params[0])
        lib = be.lib
        n, c = x.shape
        k = idx.shape[1]
        cs, T = c // 8, 3 + c + c // 8
        t = blk.transformer
        lp, lw = t.linear_p, t.linear_w
        training = blk.training
        (W1, g1, b1, Wq, bq, Wk, bk, Wv, bv, Wp1, bp1, gp, bp, Wp2, bp2, g1p, b1p, Ww1, bw1, g2p, b2p, Ww2, bw2, g2, b2, W3, g3, b3) = params
        nc = n * c
        sizes = [nc, 4 * c, nc, nc, nc, 2 * T, 2 * T, n * k * cs, nc, 4 * c, nc, 4 * c]   # z1 coef1 xq xk xv bn saved H t coef2 z3 coef3
        offs, tot = [], 0
        for sz in sizes:
            offs.append(tot)
            tot += _al(sz)
        act = torch.empty((tot,), dtype=torch.float32, device=x.device)
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        scratch = torch.empty((max(int(lib.pdf_rowlin_partial_floats(n, c)), int(lib.pdf_bn_partial_floats(n, c)),
                                   int(lib.pdf_pt_layer_partial_floats(n, k, c))),), dtype=torch.float32, device=x.device)
        base = act.data_ptr()
        ap = [base + 4 * o for o in offs]
        bn1, bn2, bn3 = blk.bn1, blk.bn2, blk.bn3
        ptrs = [x.data_ptr(), W1.data_ptr(), g1.data_ptr(), b1.data_ptr(), bn1.running_mean.data_ptr(), bn1.running_var.data_ptr(),
                Wq.data_ptr(), bq.data_ptr(), Wk.data_ptr(), bk.data_ptr(), Wv.data_ptr(), bv.data_ptr(), p.data_ptr(), idx.data_ptr(),
                Wp1.data_ptr(), bp1.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(), Ww1.data_ptr(), bw1.data_ptr(), Ww2.data_ptr(), bw2.data_ptr(),
                gp.data_ptr(), bp.data_ptr(), g1p.data_ptr(), b1p.data_ptr(), g2p.data_ptr(), b2p.data_ptr(),
                lp[1].running_mean.data_ptr(), lp[1].running_var.data_ptr(), lw[0].running_mean.data_ptr(), lw[0].running_var.data_ptr(),
                lw[3].running_mean.data_ptr(), lw[3].running_var.data_ptr(),
                g2.data_ptr(), b2.data_ptr(), bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(), W3.data_ptr(),
                g3.data_ptr(), b3.data_ptr(), bn3.running_mean.data_ptr(), bn3.running_var.data_ptr(),
                *ap, y.data_ptr(), scratch.data_ptr(), be._order_ptr(idx), be._moments_ptr(idx) if training else None]
        bf16 = int(be.storage_bf16)
        be.bottleneck_forward(n, k, c, ptrs, training, bn1.eps, bn1.momentum, bf16)
        ctx.save_for_backward(x, p, idx, act, W1, Wq, Wk, Wv, W3, Wp1, bp1, Wp2, bp2, Ww1, bw1, Ww2, bw2)
        ctx.cfg = (training, offs, k, bf16)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        training, offs, k, bf16 = ctx.cfg
        if not training:
            raise RuntimeError("fused Bottleneck: backward is implemented for training mode (batch statistics)")
        x, p, idx, act, W1, Wq, Wk, Wv, W3, Wp1, bp1, Wp2, bp2, Ww1, bw1, Ww2, bw2 = ctx.saved_tensors
        be = _be()
        lib = be.lib
        n, c = x.shape
        cs, cc, nc = c // 8, c * c, n * c
        gy = gy.contiguous()
        npre, npost, nsum = cc + 2 * c + 3 * (cc + c), cc + 4 * c, int(lib.pdf_pt_layer_bwd_sums_floats(c))
        # one buffer: [grads pre | grads post | the layer's sums]; every slot is WRITTEN by the slab reductions / column sums (no zeroing)
        o_pre, o_post = 0, _al(npre)
        o_sum = o_post + _al(npost)
        grads = torch.empty((o_sum + _al(nsum),), dtype=torch.float32, device=x.device)
        gx = torch.empty((n, c), dtype=torch.float32, device=x.device)
        from . import _native
        inv_off, inv_entry, entry_base = _native.inverse_table(idx, n)   # cached on idx by the geometry pre-pass
        ssz = [nc, nc, nc, n * k * cs, n * k * 3,
               max(int(lib.pdf_bn_partial_floats(n, c)), int(lib.pdf_rowlin_partial_floats(n, c)), int(lib.pdf_pt_layer_bwd_partial_floats(n, k, c))),
               nc, nc, n * k * cs, n * k * c, nc,
               max(int(lib.pdf_rowlin_wgrad_ws_floats(n, c, c, 5)),       # the block's five weight gradients as one grouped launch
                   int(lib.pdf_rowlin_wgrad_ws_floats(n, c, c, 3)) + int(lib.pdf_rowlin_wgrad_ws_floats(n, c, c, 1)))]   # gt da gxq G2 G3 partial | gxk gxv Wsm GR | dy | wgrad slabs
        soff, tot = [], 0
        for sz in ssz:
            soff.append(tot)
            tot += _al(sz)
        scratch = torch.empty((tot,), dtype=torch.float32, device=x.device)
        a, sb, gb = act.data_ptr(), scratch.data_ptr(), grads.data_ptr()
        A = lambda i: a + 4 * offs[i]   # z1 coef1 xq xk xv bn saved H t coef2 z3 coef3
        S = lambda i: sb + 4 * soff[i]
        ptrs = [gy.data_ptr(), x.data_ptr(), A(0), A(1), W1.data_ptr(), Wq.data_ptr(), Wk.data_ptr(), Wv.data_ptr(), p.data_ptr(), idx.data_ptr(),
                Wp1.data_ptr(), bp1.data_ptr(), Wp2.data_ptr(), bp2.data_ptr(), Ww1.data_ptr(), bw1.data_ptr(), Ww2.data_ptr(), bw2.data_ptr(),
                A(5), A(6), A(7), A(2), A(3), A(4), A(8), A(10), A(9), A(11), W3.data_ptr(),
                gx.data_ptr(), gb + 4 * o_pre, gb + 4 * o_post, gb + 4 * o_sum,
                S(0), S(1), S(2), S(6), S(7), S(3), S(4), S(5), S(8), S(9), inv_off.data_ptr(), inv_entry.data_ptr(), S(10),
                be._moments_ptr(idx) if training else None, None, None, be._order_ptr(idx), S(11)]   # (46: the table's coordinate sums; 47-48: unused, csrc/block.hip)
        be.bottleneck_backward(n, k, c, ptrs, training, entry_base, bf16)
        # the ~35 gradient views as ONE split of the buffer (a slice + view per gradient was ~100 us of host time per block)
        sizes = _bottleneck_grad_sizes(c, o_post, o_sum, grads.shape[0])
        (dW1, db1, dg1, dWq, dbq, dWk, dbk, dWv, dbv, _p0,
         dW3, db2, dg2, db3, dg3, _p1,
         beta2p, gamma2p, dbw2, dWw2, beta1p, gamma1p, dbw1, dWw1, betap, gammap, _p2, dbp2, dWp2, dbp1, dWp1, _rest) = grads.split_with_sizes(sizes)
        dW1, dWq, dWk, dWv, dW3 = dW1.view(c, c), dWq.view(c, c), dWk.view(c, c), dWv.view(c, c), dW3.view(c, c)
        dWw2, dWw1, dWp2, dWp1 = dWw2.view(cs, cs), dWw1.view(cs, c), dWp2.view(c, 3), dWp1.view(3, 3)
        qkv = [dWq, dbq, dWk, dbk, dWv, dbv]
        layer = [dWp1, dbp1, gammap, betap, dWp2, dbp2, gamma1p, beta1p, dWw1, dbw1, gamma2p, beta2p, dWw2, dbw2]
        return (None, None, None, gx, dW1, dg1, db1, *qkv, *layer, dg2, db2, dW3, dg3, db3)


_GRAD_SIZES = {}


def _bottleneck_grad_sizes(c, o_post, o_sum, total):
    """Piece sizes of _BottleneckFn.backward's gradient buffer, in buffer order (csrc/block.hip halves, then the layer's sums S1 | S2 | S3 |
    S4 of csrc/fused_layer.hip; alignment pads and the trailing scratch are pieces too)."""
    key = (c, o_post, o_sum, total)
    if key not in _GRAD_SIZES:
        cs, cc = c // 8, c * c
        pre = [cc, c, c, cc, c, cc, c, cc, c]
        post = [cc, c, c, c, c]
        sums = [cs, cs, cs, cs * cs, c, c, cs, cs * c, 3, 3, 2, c, 3 * c, 3, 9]
        sizes = pre + [o_post - sum(pre)] + post + [o_sum - o_post - sum(post)] + sums
        sizes.append(total - sum(sizes))
        assert min(sizes) >= 0
        _GRAD_SIZES[key] = sizes
    return _GRAD_SIZES[key]


def _bottleneck_params(blk):
    """The block's parameters in the order _BottleneckFn takes them.  The list of Parameter OBJECTS is cached on the block (24 nn.Module
    attribute look-ups per call otherwise: ~0.7 ms of host time per step over the 18 blocks); in-place updates (optimizer, load_state_dict,
    .to()) keep the objects, re-assigning a parameter or calling ``Module._apply`` drops the cache."""
    cached = blk.__dict__.get("_pdf_params")
    if cached is not None and cached[0] == len(blk._parameters) + len(blk._modules):
        return cached[1]
    t = blk.transformer
    lst = _bottleneck_params_uncached(blk)
    blk.__dict__["_pdf_params"] = (len(blk._parameters) + len(blk._modules), lst)
    return lst


def _bottleneck_params_uncached(blk):
    t = blk.transformer
    return [blk.linear1.weight, blk.bn1.weight, blk.bn1.bias, t.linear_q.weight, t.linear_q.bias, t.linear_k.weight, t.linear_k.bias,
            t.linear_v.weight, t.linear_v.bias, *t._param_list(), blk.bn2.weight, blk.bn2.bias, blk.linear3.weight, blk.bn3.weight, blk.bn3.bias]


def bottleneck(blk, p, x, o):
    """Bottleneck.forward through the fused kernels: one autograd node when the attention layer is fused too, else the two
    host-side halves around PointTransformerLayer.attend."""
    t = blk.transformer
    if t._fused_ok(x):
        from . import pointops
        idx, _ = pointops.knn_query(t.nsample, p, o, p, o)
        y = _BottleneckFn.apply(blk, idx, p, x, *_bottleneck_params(blk))
        if blk.training:
            lp, lw = t.linear_p, t.linear_w
            bump_counters([blk.bn1.num_batches_tracked, blk.bn2.num_batches_tracked, blk.bn3.num_batches_tracked,
                           lp[1].num_batches_tracked, lw[0].num_batches_tracked, lw[3].num_batches_tracked])
        return y
    xq, xk, xv = _BlockPre.apply(x, blk.linear1.weight, blk.bn1.weight, blk.bn1.bias, t.linear_q.weight, t.linear_q.bias,
                                 t.linear_k.weight, t.linear_k.bias, t.linear_v.weight, t.linear_v.bias, blk)
    a = t.attend(p, x, o, xq, xk, xv)
    y = _BlockPost.apply(a.contiguous(), x, blk.bn2.weight, blk.bn2.bias, blk.linear3.weight, blk.bn3.weight, blk.bn3.bias, blk)
    if blk.training:
        bump_counters([blk.bn1.num_batches_tracked, blk.bn2.num_batches_tracked, blk.bn3.num_batches_tracked])
    return y


# ------------------------------------------------------------------------------------------------------------------
# Fused TransitionDown with stride (csrc/transition_down.hip): one autograd node, one host call per direction.
# ------------------------------------------------------------------------------------------------------------------
class _TransitionDownFn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, mod, idx, rel4, Z, consts, x, W, gamma, beta):
        be = _be()
        _rcd(idx, rel4, Z, x, W)
        lib = be.lib
        n, cin = x.shape
        m, cout = idx.shape[0], W.shape[0]
        bn, training = mod.bn, mod.training
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        coef, out, gram = e(4 * cout), e(m, cout), e(int(lib.pdf_td_gram_floats(cin)))
        arg = torch.empty((m, cout), dtype=torch.uint8, device=x.device)
        ws = e(int(lib.pdf_td_fwd_scratch_floats(n, cin)))
        ptrs = [x.data_ptr(), idx.data_ptr(), rel4.data_ptr(), Z.data_ptr(), consts.data_ptr(), W.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                bn.running_mean.data_ptr(), bn.running_var.data_ptr(), coef.data_ptr(), out.data_ptr(), arg.data_ptr(), gram.data_ptr(), ws.data_ptr()]
        rc = lib.pdf_td_forward(n, m, cin, cout, (c_void_p * len(ptrs))(*ptrs), int(training), ctypes.c_float(bn.eps),
                                ctypes.c_float(bn.momentum), _mma(), be._stream())
        if rc != 0:
            raise RuntimeError(f"pdf_td_forward failed with status {rc}")
        ctx.save_for_backward(x, idx, rel4, Z, consts, W, gamma, beta, coef, out, arg, gram)
        ctx.training = training
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, gout):
        if not ctx.training:
            raise RuntimeError("fused TransitionDown: backward is implemented for training mode (batch statistics)")
        x, idx, rel4, Z, consts, W, gamma, beta, coef, out, arg, gram = ctx.saved_tensors
        be = _be()
        lib = be.lib
        n, cin = x.shape
        m, cout = out.shape
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        gx, dW, dgb = e(n, cin), e(cout, 3 + cin), e(2 * cout)
        scratch = e(int(lib.pdf_td_bwd_scratch_floats(m, cin, cout)))
        gout = gout.contiguous()
        from . import _native
        inv_off, inv_entry, entry_base = _native.inverse_table(idx, n)   # cached on idx by the geometry pre-pass: destination-order scatter
        ptrs = [gout.data_ptr(), out.data_ptr(), arg.data_ptr(), x.data_ptr(), idx.data_ptr(), rel4.data_ptr(), Z.data_ptr(), consts.data_ptr(),
                W.data_ptr(), gamma.data_ptr(), beta.data_ptr(), coef.data_ptr(), gram.data_ptr(), gx.data_ptr(), dW.data_ptr(), dgb.data_ptr(),
                scratch.data_ptr(), inv_off.data_ptr(), inv_entry.data_ptr()]
        rc = lib.pdf_td_backward(n, m, cin, cout, (c_void_p * len(ptrs))(*ptrs), int(entry_base), _mma(), be._stream())
        if rc != 0:
            raise RuntimeError(f"pdf_td_backward failed with status {rc}")
        return None, None, None, None, None, gx, dW, dgb[cout:], dgb[:cout]


def transition_down(mod, geom, level, new_level, x):
    """TransitionDown (stride > 1) through the fused kernels; the caller checked ``transition_down_ok``."""
    idx, _ = geom.knn(mod.nsample, level, new_level)
    rel4, Z, _, consts = geom.td(mod.nsample, level, new_level)
    out = _TransitionDownFn.apply(mod, idx, rel4, Z, consts, x, mod.linear.weight, mod.bn.weight, mod.bn.bias)
    if mod.training:
        bump_counters([mod.bn.num_batches_tracked])
    return out


def transition_down_ok(mod, x):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and type(mod.bn) is torch.nn.BatchNorm1d
            and mod.bn.affine and mod.bn.track_running_stats and mod.bn.momentum is not None and mod.linear.bias is None):
        return False
    if torch.is_grad_enabled() and not mod.training and any(p.requires_grad for p in mod.parameters()):
        return False
    return bool(_be().lib.pdf_td_supported(mod.nsample, x.shape[1], mod.linear.weight.shape[0]))


# ------------------------------------------------------------------------------------------------------------------
# Linear (+ bias) -> BatchNorm1d -> (ReLU) as one autograd node (csrc/block.hip: pdf_linbn_*): TransitionUp, heads.
# ------------------------------------------------------------------------------------------------------------------
class _LinBnFn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, lin, bn, relu, x, W, b, gamma, beta):
        be = _be()
        _rcd(x, W, gamma)
        lib = be.lib
        n, k = x.shape
        o = W.shape[0]
        training = bn.training
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        z, coef, y = e(n, o), e(4 * o), e(n, o)
        partial = e(int(lib.pdf_rowlin_partial_floats(n, o)))
        ptrs = [x.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None, gamma.data_ptr(), beta.data_ptr(),
                bn.running_mean.data_ptr(), bn.running_var.data_ptr(), z.data_ptr(), coef.data_ptr(), y.data_ptr(), partial.data_ptr()]
        rc = lib.pdf_linbn_forward(n, k, o, (c_void_p * len(ptrs))(*ptrs), int(training), int(relu), ctypes.c_float(bn.eps),
                                   ctypes.c_float(bn.momentum), _mma(), be._stream())
        if rc != 0:
            raise RuntimeError(f"pdf_linbn_forward failed with status {rc}")
        ctx.save_for_backward(x, z, coef, W)
        ctx.cfg = (training, relu, b is not None)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        x, z, coef, W = ctx.saved_tensors
        training, relu, has_bias = ctx.cfg
        be = _be()
        lib = be.lib
        n, k = x.shape
        o = W.shape[0]
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        need_gx = ctx.needs_input_grad[3]
        gx = e(n, k) if need_gx else None
        grads, gz = e(o * k + 3 * o), e(n, o)
        partial = e(int(lib.pdf_bn_partial_floats(n, o)))
        gy = gy.contiguous()
        ws = be.wgrad_workspace(n, k, o, 1, x.device)
        ptrs = [gy.data_ptr(), x.data_ptr(), z.data_ptr(), coef.data_ptr(), W.data_ptr(), gx.data_ptr() if need_gx else None, grads.data_ptr(),
                gz.data_ptr(), partial.data_ptr(), ws.data_ptr()]
        rc = lib.pdf_linbn_backward(n, k, o, (c_void_p * len(ptrs))(*ptrs), int(training), int(relu), _mma(), be._stream())
        if rc != 0:
            raise RuntimeError(f"pdf_linbn_backward failed with status {rc}")
        ok = o * k
        return (None, None, None, gx, grads[:ok].view(o, k), grads[ok:ok + o] if has_bias else None, grads[ok + 2 * o:ok + 3 * o], grads[ok + o:ok + 2 * o])


LINBN = True   # module-wide switch of the Linear -> BatchNorm1d -> ReLU node (tests compare both paths)


def linbn_ok(lin, bn, x):
    if not LINBN:
        return False
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and type(bn) is torch.nn.BatchNorm1d and bn.affine
            and bn.track_running_stats and bn.momentum is not None):
        return False
    k, o = lin.in_features, lin.out_features
    if k not in (32, 64, 128, 256, 512) or o % 16 or not _be().bn_supported(o):
        return False
    if torch.is_grad_enabled() and not bn.training and any(p.requires_grad for p in list(lin.parameters()) + list(bn.parameters())):
        return False  # (eval-mode backward goes through the composed ops)
    return True


def linear_bn_act(lin, bn, x, relu):
    y = _LinBnFn.apply(lin, bn, relu, x, lin.weight, lin.bias, bn.weight, bn.bias)
    if bn.training:
        bump_counters([bn.num_batches_tracked])
    return y
