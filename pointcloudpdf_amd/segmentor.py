"""Segmentor wrapper and losses of the hot path -- pointcept/models/default.py:39-62,
pointcept/models/losses/builder.py:13-27, pointcept/models/losses/misc.py:14-39 (contract only: dict in, dict out)."""
import torch
import torch.nn as nn

from . import dense
from .dense import _amp_bwd, _amp_fwd   # (custom nodes keep fp32 tensors under autocast: dense.py)
from .registry import LOSSES, MODELS, build_model


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """losses/misc.py:14-39 (ignore_index defaults to -1: unknown classes are relabelled -1 by the trainer)."""

    def __init__(self, weight=None, size_average=None, reduce=None, reduction="mean", label_smoothing=0.0,
                 loss_weight=1.0, ignore_index=-1):
        super().__init__()
        self.loss_weight = loss_weight
        w = torch.tensor(weight) if weight is not None else None
        self.loss = nn.CrossEntropyLoss(weight=w, size_average=size_average, ignore_index=ignore_index, reduce=reduce,
                                        reduction=reduction, label_smoothing=label_smoothing)

    def forward(self, pred, target):
        l = self.loss
        if (pred.is_cuda and pred.dim() == 2 and pred.dtype == torch.float32 and pred.shape[1] <= 64 and target.dtype == torch.int64
                and l.weight is None and l.label_smoothing == 0.0 and l.reduction == "mean"):
            return _FusedCE.apply(pred.contiguous(), target.contiguous(), int(l.ignore_index)) * self.loss_weight
        return l(pred, target) * self.loss_weight


class _FusedCE(torch.autograd.Function):
    """nn.CrossEntropyLoss(mean, ignore_index) over (N, C <= 64) fp32 logits as one HIP kernel per direction (csrc/loss.hip)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, pred, target, ignore):
        import ctypes
        from . import _native

        be = _native.hip_backend()
        n, c = pred.shape
        grad = torch.empty_like(pred)
        # [sum, count, mean, -] + the per-workgroup partial sums (added in a fixed order: csrc/loss.hip)
        acc = torch.empty((int(be.lib.pdf_ce_workspace_floats()),), dtype=torch.float32, device=pred.device)
        _native.require_current_device(pred, target)
        s = ctypes.c_void_p(_native.raw_stream())
        rc = be.lib.pdf_ce_forward(n, c, pred.data_ptr(), target.data_ptr(), ignore, grad.data_ptr(), acc.data_ptr(), acc.data_ptr() + 8, s)
        if rc != 0:
            raise RuntimeError(f"pdf_ce_forward failed with status {rc}")
        ctx.save_for_backward(grad, acc)
        return acc[2]

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        import ctypes
        from . import _native

        dlogits, acc = ctx.saved_tensors
        n, c = dlogits.shape
        gy = gy.contiguous().float()
        out = torch.empty_like(dlogits)   # the saved buffer stays untouched: the node may be differentiated again (retain_graph)
        _native.require_current_device(dlogits, gy)
        s = ctypes.c_void_p(_native.raw_stream())
        rc = _native.hip_backend().lib.pdf_ce_backward(n, c, dlogits.data_ptr(), acc.data_ptr(), gy.data_ptr(), out.data_ptr(), s)
        if rc != 0:
            raise RuntimeError(f"pdf_ce_backward failed with status {rc}")
        return out, None, None


class Criteria:
    """losses/builder.py:13-27: sum of the configured losses; an empty list returns the prediction itself."""

    def __init__(self, cfg=None):
        self.cfg = cfg if cfg is not None else []
        self.criteria = [LOSSES.build(cfg=c) for c in self.cfg]

    def __call__(self, pred, target):
        if len(self.criteria) == 0:
            return pred
        loss = 0
        for c in self.criteria:
            loss = loss + c(pred, target)
        return loss


def build_criteria(cfg):
    return Criteria(cfg)


@MODELS.register_module()
class DefaultSegmentor(nn.Module):
    """default.py:39-62: train -> {loss}; eval with labels -> {loss, seg_logits}; test -> {seg_logits}."""

    def __init__(self, backbone=None, criteria=None):
        super().__init__()
        self.backbone = build_model(backbone)
        self.criteria = build_criteria(criteria)

    @dense.fp32_path   # (the loss as well: fp32 tensors, reduced-precision product operands only -- dense.fp32_path)
    def forward(self, input_dict):
        if "condition" in input_dict.keys():
            input_dict["condition"] = input_dict["condition"][0]
        seg_logits = self.backbone(input_dict)
        if self.training:
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]))
        if "segment" in input_dict.keys():
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]), seg_logits=seg_logits)
        return dict(seg_logits=seg_logits)
