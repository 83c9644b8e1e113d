"""PDF open-world recognizers on the hot path.

* ``PTRecognizer`` -- the PDF **U-decoder** uncertainty head, pointcept/recognizers/recognizer_model/pt_v1.py:8-44
  (registered as ``PointTransformer-Recognizer``; parameter names ``dec{1..5}.linear{1,2}``, ``confidence``).
* ``PointPdfV1`` -- forward/score/loss part of pointcept/recognizers/ours/pointpdf_v1m1_base.py:72-116 and
  ``trigger_operation`` (:384-398).  The pseudo-label pass (:118-382; third-party ball query + CPU graph code) is
  a "next" row of the scope table: a ``pseudo_mask_fn(coord, seg_logits, offset) -> bool mask`` can be plugged in.
* ``MaxProbability`` -- MSP / max-logit baselines, pointcept/recognizers/max_probability/max_probability_v1m1_base.py:7-32.
"""
import os

import torch
import torch.nn as nn

from . import dense
from .point_transformer import TransitionUp, _seq
from .registry import MODELS, RECOGNIZER, build_model
from .segmentor import build_criteria


@MODELS.register_module("PointTransformer-Recognizer")
class PTRecognizer(nn.Module):
    def __init__(self):
        super().__init__()
        planes = [32, 64, 128, 256, 512]
        self.dec5 = TransitionUp(planes[4], planes[4])
        self.dec4 = TransitionUp(planes[4], planes[3])
        self.dec3 = TransitionUp(planes[3], planes[2])
        self.dec2 = TransitionUp(planes[2], planes[1])
        self.dec1 = TransitionUp(planes[1], planes[0])
        self.confidence = nn.Sequential(
            nn.Linear(planes[0], planes[0]), nn.BatchNorm1d(planes[0]), nn.ReLU(inplace=True), nn.Linear(planes[0], 1)
        )

    @dense.fp32_path
    def forward(self, model_hooks):
        enc = [model_hooks[f"backbone.enc{i}"]["forward_output"] for i in range(1, 6)]
        dec = [model_hooks[f"backbone.dec{i}.1"]["forward_output"][1] for i in range(1, 6)]
        (p1, _, o1), (p2, _, o2), (p3, _, o3), (p4, _, o4), (p5, x5_enc, o5) = enc
        x1, x2, x3, x4, x5_dec = dec
        r5 = self.dec5([p5, x5_dec, o5], [p5, x5_enc, o5])
        r4 = self.dec4([p4, x4, o4], [p5, r5, o5])
        r3 = self.dec3([p3, x3, o3], [p4, r4, o4])
        r2 = self.dec2([p2, x2, o2], [p3, r3, o3])
        r1 = self.dec1([p1, x1, o1], [p2, r2, o2])
        return _seq(self.confidence, r1)  # (n, 1)


@RECOGNIZER.register_module("PointPdf-v1m1")
class PointPdfV1(nn.Module):
    def __init__(self, recognizer, criteria, loss_weight, step_loss_weight, num_classes, start_epoch,
                 kp_ball_radius=None, kp_max_neighbor=None, condition_from=None, beta=None, seed_from=None,
                 seed_range=None, num_seed=None, slide_window=False, adaptive_radius=False, softmax_score=True,
                 pseudo_mask_fn=None):
        super().__init__()
        self.need_input = True
        self.init_disable_update = True
        self.start_epoch = start_epoch
        self.runtime_update = True
        self.num_classes = num_classes
        self.alpha = loss_weight
        self.step_loss_weight = step_loss_weight
        self.kp_ball_radius, self.kp_max_neighbor = kp_ball_radius, kp_max_neighbor
        self.recognizer = build_model(recognizer)
        self.criteria = build_criteria(criteria)
        self.condition_from, self.beta = condition_from, beta
        self.seed_from, self.seed_range, self.num_seed = seed_from, seed_range, num_seed
        self.slide_window, self.adaptive_radius = slide_window, adaptive_radius
        self.softmax_score = softmax_score
        self.pseudo_mask_fn = pseudo_mask_fn
        # (off by default: measured inconclusive on a loaded pod -- 38.7 vs 42.1 ms per step averaged over three traced runs each, 43.6 vs
        # 42.1 over four untraced ones, profiles/r04_pl_side_stream_ab.txt; PDFOPS_PL_SIDE_STREAM=1 turns it on)
        self.pass_on_side_stream = os.environ.get("PDFOPS_PL_SIDE_STREAM", "0") == "1"
        self._logits_ready, self._pass_stream = None, None
        self.model_hooks = None
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def get_pseudo_mask(self, coord, seg_logits, offset, offset_host=None, geometry=None):
        if self.pseudo_mask_fn is None:
            raise NotImplementedError(
                "PDF pseudo-label pass (pointpdf_v1m1_base.py:118-382) is row f-2 of the scope table; "
                "pass pseudo_mask_fn=... to supply the mask"
            )
        with torch.no_grad():
            ready = getattr(self, "_logits_ready", None)
            self._logits_ready = None
            extra = {}
            if geometry is not None and getattr(self.pseudo_mask_fn, "accepts_geometry", False):
                extra["geometry"] = geometry      # (its radius table, when the coordinate pre-pass made one)
            if ready is None or not self.pass_on_side_stream:
                if offset_host is not None and getattr(self.pseudo_mask_fn, "accepts_offset_host", False):
                    return self.pseudo_mask_fn(coord, seg_logits, offset, offset_host=offset_host, **extra).bool()
                return self.pseudo_mask_fn(coord, seg_logits, offset, **extra).bool()
            # The pass only needs the segmentor's logits, but it is issued after the U-decoder's forward (upstream's order, :84-93): on
            # the same stream its first host read would wait for the decoder too.  On a side stream that waits for the logits alone the
            # decoder's forward runs on the device while the host walks through the pass.
            cur = torch.cuda.current_stream()
            if self._pass_stream is None:
                self._pass_stream = torch.cuda.Stream(device=coord.device)
            self._pass_stream.wait_event(ready)
            with torch.cuda.stream(self._pass_stream):
                mask = self.pseudo_mask_fn(coord, seg_logits, offset, **extra).bool()
            cur.wait_stream(self._pass_stream)
            mask.record_stream(cur)
            return mask

    def trigger_operation(self):
        """pointpdf_v1m1_base.py:384-398: freeze the U-decoder until start_epoch, then release; decay alpha once."""
        if self.init_disable_update:
            for p in self.recognizer.parameters():
                p.requires_grad = False
            self.init_disable_update = False
        if self.epoch >= self.start_epoch and self.runtime_update:
            for p in self.recognizer.parameters():
                p.requires_grad = True
            self.runtime_update = False
        if self.epoch > self.start_epoch + 1 and self.step_loss_weight:
            self.alpha = self.alpha * 0.1
            self.step_loss_weight = False

    @dense.fp32_path
    def forward(self, input_dict):
        seg_logits = self.model_hooks["backbone"]["forward_output"]
        self.trigger_operation()
        if (self.training and self.pseudo_mask_fn is not None and self.pass_on_side_stream and seg_logits.is_cuda
                and self.epoch >= self.start_epoch and not torch.cuda.is_current_stream_capturing()):
            self._logits_ready = torch.cuda.Event()
            self._logits_ready.record()          # (before the U-decoder's forward is issued: see get_pseudo_mask)
        score = self.recognizer(self.model_hooks)
        if self.training:
            if self.epoch < self.start_epoch:
                return dict(score=score)
            pseudo_mask = self.get_pseudo_mask(input_dict["coord"], seg_logits, input_dict["offset"], input_dict.get("offset_host"),
                                               input_dict.get("pdf_geometry"))
            # segment_pseudo[pseudo_mask] = num_classes (pointpdf_v1m1_base.py:94-95) as a select: no index list, no host sync
            segment = input_dict["segment"]
            segment_pseudo = torch.where(pseudo_mask, segment.new_full((), self.num_classes), segment)
            full = torch.cat([seg_logits, score], -1)
            loss = self.criteria(full, segment_pseudo) * self.alpha
            if self.softmax_score:
                score = full.softmax(-1)[:, -1]
            return dict(score=score, loss=loss)
        if "segment" in input_dict.keys():
            if self.softmax_score:
                score = torch.cat([seg_logits, score], -1).softmax(-1)[:, -1]
            return dict(score=score)
        return dict(seg_logits=seg_logits)


@RECOGNIZER.register_module()
class MaxProbability:
    def __init__(self, method=None):
        if method == "msp":
            self.prob_func = self.msp
        elif method == "max_logits":
            self.prob_func = self.ml
        else:
            raise ValueError(f"Unknown MaxProbability method {method}")
        self.model_hooks = None

    def __call__(self, input_dict):
        seg_logits = self.model_hooks["backbone"]["forward_output"]
        return dict(score=-self.prob_func(seg_logits))

    @staticmethod
    def msp(seg_logits):
        return seg_logits.log_softmax(dim=-1).max(dim=-1)[0]

    @staticmethod
    def ml(seg_logits):
        return seg_logits.max(dim=-1)[0]

    def set_epoch(self, epoch):
        self.epoch = epoch
