"""Segmentor wrapper and losses of the hot path -- pointcept/models/default.py:39-62,
pointcept/models/losses/builder.py:13-27, pointcept/models/losses/misc.py:14-39 (contract only: dict in, dict out)."""
import torch
import torch.nn as nn

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
        return self.loss(pred, target) * self.loss_weight


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

    def forward(self, input_dict):
        if "condition" in input_dict.keys():
            input_dict["condition"] = input_dict["condition"][0]
        seg_logits = self.backbone(input_dict)
        if self.training:
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]))
        if "segment" in input_dict.keys():
            return dict(loss=self.criteria(seg_logits, input_dict["segment"]), seg_logits=seg_logits)
        return dict(seg_logits=seg_logits)
