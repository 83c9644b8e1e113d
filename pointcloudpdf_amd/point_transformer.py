"""PointTransformer-V1 segmentation backbone for MI355X.

Plugin-compatible with pointcept/models/point_transformer/point_transformer_seg.py:19-327: the classes are
registered as ``PointTransformer-Seg26/38/50`` with the same constructor kwargs, the same submodule tree
(``enc1..enc5`` / ``dec1..dec5`` as ``nn.Sequential``; ``linear1, bn1, transformer.{linear_q, linear_k, linear_v,
linear_p, linear_w}, bn2, linear3, bn3``; ``TransitionDown.{linear, bn}``; ``TransitionUp.{linear1, linear2}``;
``cls``) and therefore the same ``state_dict`` keys, and every stage returns ``[p, x, o]`` so that
``ModelHook`` names such as ``backbone.enc3`` / ``backbone.dec2.1`` resolve.

What differs is the execution plan:
  * all coordinate-only work (FPS, kNN, interpolation tables) comes from one ``Geometry`` per batch
    (identical results, computed once instead of 4 + 31 times) -- geometry.py;
  * no ``.item()`` / host syncs inside the layers, no (n, ns, c) transpose copies for the BatchNorm-as-LayerNorm;
  * gathers, relative coordinates, masks and concatenations run in the fused HIP gather kernels.
"""
import os

import torch

import torch.nn as nn
import torch.nn.functional as F

from . import _native, pointops
from . import dense
from .dense import _amp_bwd, _amp_fwd, bn_act as _bn_act, linear as _lin   # (custom nodes keep fp32 tensors under autocast: dense.py)
from .geometry import Geometry, tag_of
from .registry import MODELS


def _seq(seq, x):
    """Run an nn.Sequential of Linear / BatchNorm1d / ReLU members: Linear through the split-K weight-gradient path,
    BatchNorm1d (+ a directly following ReLU) through the fused normalisation kernels."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if (isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) and x.dim() == 2
                and dense.linbn_ok(m, mods[i + 1], x)):   # Linear -> BatchNorm1d (-> ReLU) as one node (csrc/block.hip)
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = dense.linear_bn_act(m, mods[i + 1], x, relu)
            i += 3 if relu else 2
            continue
        if isinstance(m, nn.Linear):
            x = _lin(m, x)
        elif isinstance(m, nn.BatchNorm1d) and type(m) is nn.BatchNorm1d and x.dim() == 2:
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = _bn_act(m, x, None, relu)
            i += 1 if relu else 0
        else:
            x = m(x)
        i += 1
    return x


class LayerNorm1d(nn.BatchNorm1d):
    """BatchNorm1d over the channel (last) dim of an (n, ns, c) tensor -- pointcept/models/point_transformer/utils.py:7-14.
    The reference transposes to (n, c, ns) and back (two full copies); statistics over all n*ns rows per channel
    are the same as BatchNorm over the (n*ns, c) view, which needs no copy."""

    def forward(self, input):
        shape = input.shape
        x2d = input.reshape(-1, shape[-1])
        if input.dtype == torch.float32 and _native.hip_backend().bn_supported(shape[-1]):
            return _bn_act(self, x2d.contiguous(), None, False).view(shape)
        return nn.BatchNorm1d.forward(self, x2d).view(shape)  # 3-channel norm of linear_p / odd widths


class _FusedPTLayer(torch.autograd.Function):
    """The whole PointTransformerLayer after the q/k/v projections as one autograd node over the fused HIP passes
    (csrc/fused_layer.hip).  Forward: no (n, ns, c) tensor is materialised (only ``H (n, ns, c/8)`` is saved).  Backward: the pass
    that forms the gradient rows ``g_r (n, ns, c)`` (B3) streams them out once and a segmented gather over the inverse kNN table
    (``sg::k_seg_rows``) sums them per source point (``g_xk``; no atomics, fixed order).  That write + re-read of ``GR`` is the one
    (n, ns, c) tensor the layer's backward still moves through HBM: 0.46 ms of the 15.2 ms step over the 18 layers (205 MB each way at
    level 1), NOT the bulk of the Bottleneck backward's traffic -- that is the passes' re-gathering of rows (DESIGN.md sections 5a, 6).
    Forming ``g_xk`` inside the segmented walk instead means redoing B3's per-entry chain (relative-position MLP, BatchNorm + ReLU masks,
    the (c/8 x c) product) in destination order with a gathered ``x_q`` row per entry: the same row volume as the ``GR`` read it would
    save, plus the arithmetic a second time (priced, not built)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, layer, idx, p, xq, xk, xv, *params):
        be = _native.hip_backend()
        lp, lw = layer.linear_p, layer.linear_w
        weights = [lp[0].weight, lp[0].bias, lp[3].weight, lp[3].bias, lw[2].weight, lw[2].bias, lw[5].weight, lw[5].bias]
        norms = [lp[1], lw[0], lw[3]]
        bn_params = [t for n in norms for t in (n.weight, n.bias)]
        training = layer.training
        bn_buffers = [t for n in norms for t in (n.running_mean, n.running_var)]
        weights = [w.detach().contiguous() for w in weights]
        bn_params = [w.detach().contiguous() for w in bn_params]
        out, bn, saved, H = be.pt_layer_forward(xq.contiguous(), xk.contiguous(), xv.contiguous(), p, idx, weights,
                                                bn_params, bn_buffers, training, norms[0].eps, norms[0].momentum)
        if training:
            dense.bump_counters([n.num_batches_tracked for n in norms])
        ctx.save_for_backward(xq, xk, xv, p, idx, bn, saved, H, *weights)
        ctx.training = training
        ctx.bf16 = int(be.storage_bf16)   # H was written in this format
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, gout):
        if not ctx.training:
            raise RuntimeError("fused PointTransformerLayer: backward is implemented for training mode (batch statistics)")
        xq, xk, xv, p, idx, bn, saved, H, *weights = ctx.saved_tensors
        be = _native.hip_backend()
        gxq, gxk, gxv, g = be.pt_layer_backward(xq.contiguous(), xk.contiguous(), xv.contiguous(), p, idx, list(weights),
                                                bn, saved, H, gout.contiguous(), storage_bf16=ctx.bf16)
        # order of *params in forward(): Wp1 bp1 gamma_p beta_p Wp2 bp2 gamma_1 beta_1 Ww1 bw1 gamma_2 beta_2 Ww2 bw2
        grads = (g["Wp1"], g["bp1"], g["gammap"], g["betap"], g["Wp2"], g["bp2"], g["gamma1"], g["beta1"], g["Ww1"],
                 g["bw1"], g["gamma2"], g["beta2"], g["Ww2"], g["bw2"])
        return (None, None, None, gxq, gxk, gxv) + grads


class PointTransformerLayer(nn.Module):
    """Vector attention with shared planes -- point_transformer_seg.py:19-78."""

    def __init__(self, in_planes, out_planes, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes = out_planes
        self.share_planes = share_planes
        self.nsample = nsample
        self.linear_q = nn.Linear(in_planes, mid_planes)
        self.linear_k = nn.Linear(in_planes, mid_planes)
        self.linear_v = nn.Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(
            nn.Linear(3, 3), LayerNorm1d(3), nn.ReLU(inplace=True), nn.Linear(3, out_planes)
        )
        self.linear_w = nn.Sequential(
            LayerNorm1d(mid_planes),
            nn.ReLU(inplace=True),
            nn.Linear(mid_planes, out_planes // share_planes),
            LayerNorm1d(out_planes // share_planes),
            nn.ReLU(inplace=True),
            nn.Linear(out_planes // share_planes, out_planes // share_planes),
        )
        self.softmax = nn.Softmax(dim=1)

    fused = True  # class-wide switch: use the fused HIP passes where they apply

    def _fused_ok(self, x):
        if not (self.fused and x.is_cuda and x.dtype == torch.float32 and self.mid_planes == self.out_planes
                and self.share_planes == 8):
            return False
        if torch.is_grad_enabled() and not self.training and any(p.requires_grad for p in self.parameters()):
            return False  # eval-mode backward is not implemented in the fused path
        if any(n.momentum is None for n in (self.linear_p[1], self.linear_w[0], self.linear_w[3])):
            return False  # cumulative-average running statistics: torch's BatchNorm
        return _native.hip_backend().pt_layer_supported(self.nsample, self.out_planes)

    def _param_list(self):
        lp, lw = self.linear_p, self.linear_w
        return [lp[0].weight, lp[0].bias, lp[1].weight, lp[1].bias, lp[3].weight, lp[3].bias, lw[0].weight, lw[0].bias,
                lw[2].weight, lw[2].bias, lw[3].weight, lw[3].bias, lw[5].weight, lw[5].bias]


    def forward(self, pxo):
        p, x, o = pxo  # (n, 3), (n, c), (b)
        x_q, x_k, x_v = _lin(self.linear_q, x), _lin(self.linear_k, x), _lin(self.linear_v, x)
        return self.attend(p, x, o, x_q, x_k, x_v)

    def attend(self, p, x, o, x_q, x_k, x_v):
        """Everything after the q/k/v projections (``x`` is only consulted for device / dtype)."""
        if self._fused_ok(x):
            idx, _ = pointops.knn_query(self.nsample, p, o, p, o)
            return _FusedPTLayer.apply(self, idx, p, x_q, x_k, x_v, *self._param_list())
        x_k, idx = pointops.knn_query_and_group(x_k, p, o, new_xyz=p, new_offset=o, nsample=self.nsample, with_xyz=True)
        x_v, _ = pointops.knn_query_and_group(x_v, p, o, new_xyz=p, new_offset=o, idx=idx, nsample=self.nsample, with_xyz=False)
        p_r, x_k = x_k[:, :, 0:3], x_k[:, :, 3:]
        p_r = self.linear_p(p_r)  # (n, ns, c); out_planes == mid_planes so the (i j) reduction upstream is the identity
        r_qk = x_k - x_q.unsqueeze(1) + p_r
        w = self.softmax(self.linear_w(r_qk))  # (n, ns, c/s), softmax over the neighbour dim
        n, ns, c = x_v.shape
        s = self.share_planes
        x = ((x_v + p_r).view(n, ns, s, c // s) * w.unsqueeze(2)).sum(1)  # einsum "n t s i, n t i -> n s i"
        return x.reshape(n, c)


class TransitionDown(nn.Module):
    """point_transformer_seg.py:81-119"""

    fused = os.environ.get("PDFOPS_FUSED_TD", "1") != "0"   # class-wide switch: csrc/transition_down.hip where it applies

    def __init__(self, in_planes, out_planes, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        if stride != 1:
            self.linear = nn.Linear(3 + in_planes, out_planes, bias=False)
            self.pool = nn.MaxPool1d(nsample)
        else:
            self.linear = nn.Linear(in_planes, out_planes, bias=False)
        self.bn = nn.BatchNorm1d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def _downsample(self, p, o):
        tag = tag_of(p)
        if tag is not None:
            geom, level = tag
            new_level, _ = geom.down(level, self.stride)
            return geom.coord(new_level), geom.offset(new_level)
        # untagged coordinates: the reference's own host-side offset arithmetic (:96-100)
        ends = [int(v) for v in o.detach().cpu().tolist()]
        n_o, count, prev = [], 0, 0
        for e in ends:
            count += (e - prev) // self.stride
            prev = e
            n_o.append(count)
        n_o = torch.tensor(n_o, dtype=torch.int32, device=p.device)
        idx = pointops.farthest_point_sampling(p, o, n_o)
        return p[idx.long(), :].contiguous(), n_o

    def _linear_bn_pool(self, x):
        """(m, ns, 3 + c) grouped rows -> Linear -> BatchNorm -> ReLU -> max over the neighbours (point_transformer_seg.py:112-117)."""
        m, ns = x.shape[0], x.shape[1]
        y = _bn_act(self.bn, _lin(self.linear, x.view(m * ns, -1)), None, True)  # BN over all m*ns rows == BN1d on (m, c, ns)
        return self.pool(y.view(m, ns, -1).transpose(1, 2)).squeeze(-1)  # (m, c)

    def forward(self, pxo):
        p, x, o = pxo  # (n, 3), (n, c), (b)
        if self.stride != 1:
            n_p, n_o = self._downsample(p, o)
            tag, ntag = tag_of(p), tag_of(n_p)
            if self.fused and tag is not None and ntag is not None and tag[0] is ntag[0] and dense.transition_down_ok(self, x):
                return [n_p, dense.transition_down(self, tag[0], tag[1], ntag[1], x), n_o]
            x, _ = pointops.knn_query_and_group(x, p, offset=o, new_xyz=n_p, new_offset=n_o, nsample=self.nsample, with_xyz=True)
            x = self._linear_bn_pool(x)
            p, o = n_p, n_o
        else:
            x = _bn_act(self.bn, _lin(self.linear, x), None, True)  # (n, c)
        return [p, x, o]


def _scene_rows_kernels(x):
    """The backend's per-scene row kernels (csrc/scene_rows.hip) for float32 device rows, else None (host tensors: the torch composition)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0):
        return None
    if os.environ.get("PDFOPS_HEAD_TORCH_SUMS") == "1":   # rounds 1-4: torch's sum(0) per scene (tools/probes/replay_reduction_probe.py shows what
        return None                                        # that does to replays of a captured step from 512 rows per scene on)
    be = _native.backend_for(x)
    return be if "scene_sum_rows" in getattr(be, "_fn", {}) else None


class _SceneMean(torch.autograd.Function):
    """``x (N, c)`` -> the mean row of every scene (b, c) (point_transformer_seg.py:152-154: ``x_b.sum(0, True) / cnt``).  One launch with
    a fixed summation order each way; NOT torch's ``sum(0)``: from 512 rows on that is a multi-workgroup reduction whose semaphore is cleared
    by a memset node when the step is captured, and replays then return garbage on this stack (csrc/scene_rows.hip)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, offset, sizes_host):
        be = _scene_rows_kernels(x)
        ctx.sizes, ctx.n = [int(v) for v in sizes_host], x.shape[0]
        if be is None:
            ctx.offset = None
            return torch.cat([ch.sum(0, True) / ch.shape[0] for ch in x.split(ctx.sizes, dim=0)], 0)
        x = x.contiguous()
        ctx.offset = offset = offset.int().contiguous()
        out = torch.empty((len(ctx.sizes), x.shape[1]), dtype=x.dtype, device=x.device)
        be._call("scene_sum_rows", len(ctx.sizes), offset, x.shape[1], x, x.shape[1], 1, out)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        if ctx.offset is None:
            return torch.cat([(g[i:i + 1] / n).expand(n, -1) for i, n in enumerate(ctx.sizes)], 0), None, None
        g = g.contiguous()
        out = torch.empty((ctx.n, g.shape[1]), dtype=g.dtype, device=g.device)
        _native.backend_for(g)._call("scene_repeat_rows", len(ctx.sizes), ctx.offset, ctx.n, g.shape[1], g, 1, out)
        return out, None, None


class _RowsPerScene(torch.autograd.Function):
    """``ctx (b, c)`` repeated over the points of its scene -> (N, c) (point_transformer_seg.py:155-158: ``x_b.repeat(cnt, 1)``).
    torch.repeat_interleave's backward is an index_add_ with float atomics: the per-scene sum of the incoming rows then depends on the
    arrival order (found by the bit-reproducibility check at 2 x 4,500 points).  Here the backward is a plain per-scene sum in a fixed
    order (csrc/scene_rows.hip; see _SceneMean for why it is not torch's ``sum(0)``)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, rows, offset, sizes_dev, sizes_host, total):
        ctx.sizes = [int(v) for v in sizes_host]
        be = _scene_rows_kernels(rows)
        if be is None:
            ctx.offset = None
            return torch.repeat_interleave(rows, sizes_dev, dim=0, output_size=total)
        rows = rows.contiguous()
        ctx.offset = offset = offset.int().contiguous()
        out = torch.empty((total, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        be._call("scene_repeat_rows", len(ctx.sizes), offset, total, rows.shape[1], rows, 0, out)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        if ctx.offset is None:
            return torch.cat([ch.sum(0, keepdim=True) for ch in g.split(ctx.sizes, dim=0)], 0), None, None, None, None
        g = g.contiguous()
        out = torch.empty((len(ctx.sizes), g.shape[1]), dtype=g.dtype, device=g.device)
        _native.backend_for(g)._call("scene_sum_rows", len(ctx.sizes), ctx.offset, g.shape[1], g, g.shape[1], 0, out)
        return out, None, None, None, None


class TransitionUp(nn.Module):
    """point_transformer_seg.py:122-168"""

    def __init__(self, in_planes, out_planes=None):
        super().__init__()
        if out_planes is None:
            self.linear1 = nn.Sequential(nn.Linear(2 * in_planes, in_planes), nn.BatchNorm1d(in_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, in_planes), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(nn.Linear(out_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))

    @staticmethod
    def _scene_sizes(p, o):
        tag = tag_of(p)
        ends = tag[0].offset_host(tag[1]) if tag is not None else [int(v) for v in o.detach().cpu().tolist()]
        return [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]

    def forward(self, pxo1, pxo2=None):
        if pxo2 is None:
            p, x, o = pxo1  # head: append the scene-mean context to every point (:148-161)
            sizes = self._scene_sizes(p, o)
            means = _SceneMean.apply(x, o, sizes)  # (b, c)
            ctx = _seq(self.linear2, means)
            tag = tag_of(p)
            sizes_dev = tag[0].sizes(tag[1]) if tag is not None else torch.diff(o.long(), prepend=o.new_zeros(1).long())
            rep = _RowsPerScene.apply(ctx, o, sizes_dev, sizes, x.shape[0])
            x = _seq(self.linear1, torch.cat((x, rep), 1))
        else:
            p1, x1, o1 = pxo1
            p2, x2, o2 = pxo2
            x = _seq(self.linear1, x1) + pointops.interpolation(p2, p1, _seq(self.linear2, x2), o2, o1)
        return x


class Bottleneck(nn.Module):
    """point_transformer_seg.py:171-192"""

    expansion = 1

    def __init__(self, in_planes, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = nn.Linear(in_planes, planes, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.transformer = PointTransformerLayer(planes, planes, share_planes, nsample)
        self.bn2 = nn.BatchNorm1d(planes)
        self.linear3 = nn.Linear(planes, planes * self.expansion, bias=False)
        self.bn3 = nn.BatchNorm1d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)

    def _apply(self, fn, *a, **kw):   # (.to() / .cuda() / .float() may replace Parameter objects: drop dense._bottleneck_params' cache)
        self.__dict__.pop("_pdf_params", None)
        return super()._apply(fn, *a, **kw)

    matrix_core = os.environ.get("PDFOPS_MATRIX_CORE", "1") != "0"  # class-wide switch: Linear + BatchNorm chains through csrc/rowlin.hip

    def forward(self, pxo):
        p, x, o = pxo
        identity = x
        t = self.transformer
        if self.matrix_core and dense.fused_ok(x, self.bn1, self.bn2, self.bn3):
            return [p, dense.bottleneck(self, p, x, o), o]
        x = _bn_act(self.bn1, _lin(self.linear1, x), None, True)
        x = _bn_act(self.bn2, self.transformer([p, x, o]), None, True)
        x = _bn_act(self.bn3, _lin(self.linear3, x), identity, True)  # relu(bn3(.) + identity)
        return [p, x, o]


class PointTransformerSeg(nn.Module):
    """point_transformer_seg.py:195-303"""

    def __init__(self, block, blocks, in_channels=6, num_classes=13):
        super().__init__()
        self.in_channels = in_channels
        self.in_planes, planes = in_channels, [32, 64, 128, 256, 512]
        share_planes = 8
        stride, nsample = [1, 4, 4, 4, 4], [8, 16, 16, 16, 16]
        self.strides, self.nsamples = stride, nsample
        for i in range(5):  # enc1..enc5: N/1, N/4, N/16, N/64, N/256
            setattr(self, f"enc{i + 1}", self._make_enc(block, planes[i], blocks[i], share_planes, stride[i], nsample[i]))
        for i in range(4, -1, -1):  # dec5 (head: transform p5) .. dec1 (fusion p2 and p1)
            setattr(self, f"dec{i + 1}", self._make_dec(block, planes[i], 1, share_planes, nsample[i], is_head=(i == 4)))
        self.cls = nn.Sequential(
            nn.Linear(planes[0], planes[0]), nn.BatchNorm1d(planes[0]), nn.ReLU(inplace=True), nn.Linear(planes[0], num_classes)
        )
        self._last_geometry = None

    def _make_enc(self, block, planes, blocks, share_planes=8, stride=1, nsample=16):
        layers = [TransitionDown(self.in_planes, planes * block.expansion, stride, nsample)]
        self.in_planes = planes * block.expansion
        layers += [block(self.in_planes, self.in_planes, share_planes, nsample=nsample) for _ in range(blocks)]
        return nn.Sequential(*layers)

    def _make_dec(self, block, planes, blocks, share_planes=8, nsample=16, is_head=False):
        layers = [TransitionUp(self.in_planes, None if is_head else planes * block.expansion)]
        self.in_planes = planes * block.expansion
        layers += [block(self.in_planes, self.in_planes, share_planes, nsample=nsample) for _ in range(blocks)]
        return nn.Sequential(*layers)

    def geometry_for(self, data_dict):
        """The batch's Geometry: reuse a prefetched one (``data_dict['pdf_geometry']``) or build it here."""
        geom = data_dict.get("pdf_geometry") if isinstance(data_dict, dict) else None
        if geom is None:
            geom = Geometry(data_dict["coord"], data_dict["offset"], data_dict.get("offset_host"))
            try:
                data_dict["pdf_geometry"] = geom  # keeps the memo table alive for the recognizer pass
            except TypeError:
                pass
        self._last_geometry = geom
        return geom

    @dense.fp32_path
    def forward(self, data_dict):
        geom = self.geometry_for(data_dict)
        p0, o0 = geom.coord(0), geom.offset(0)
        x0 = data_dict["feat"]
        if x0.dtype in (torch.float16, torch.bfloat16):   # (an autocast producer upstream: the path itself is fp32)
            x0 = x0.float()
        p1, x1, o1 = self.enc1([p0, x0, o0])
        p2, x2, o2 = self.enc2([p1, x1, o1])
        p3, x3, o3 = self.enc3([p2, x2, o2])
        p4, x4, o4 = self.enc4([p3, x3, o3])
        p5, x5, o5 = self.enc5([p4, x4, o4])
        x5 = self.dec5[1:]([p5, self.dec5[0]([p5, x5, o5]), o5])[1]
        x4 = self.dec4[1:]([p4, self.dec4[0]([p4, x4, o4], [p5, x5, o5]), o4])[1]
        x3 = self.dec3[1:]([p3, self.dec3[0]([p3, x3, o3], [p4, x4, o4]), o3])[1]
        x2 = self.dec2[1:]([p2, self.dec2[0]([p2, x2, o2], [p3, x3, o3]), o2])[1]
        x1 = self.dec1[1:]([p1, self.dec1[0]([p1, x1, o1], [p2, x2, o2]), o1])[1]
        return _seq(self.cls, x1)


@MODELS.register_module("PointTransformer-Seg26")
class PointTransformerSeg26(PointTransformerSeg):
    def __init__(self, **kwargs):
        super().__init__(Bottleneck, [1, 1, 1, 1, 1], **kwargs)


@MODELS.register_module("PointTransformer-Seg38")
class PointTransformerSeg38(PointTransformerSeg):
    def __init__(self, **kwargs):
        super().__init__(Bottleneck, [1, 2, 2, 2, 2], **kwargs)


@MODELS.register_module("PointTransformer-Seg50")
class PointTransformerSeg50(PointTransformerSeg):
    def __init__(self, **kwargs):
        super().__init__(Bottleneck, [1, 2, 3, 5, 2], **kwargs)
