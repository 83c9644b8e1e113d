"""Open-world evaluation metrics on the device (second half): the histogram IoU of
pointcept/utils/misc.py:55-67 (``intersection_and_union_gpu``), the known-class summary of ``OpenSegEvaluator.eval``
(engines/hooks/evaluator.py:77-86) and its per-batch AUPR / AUROC bookkeeping (:196-216, utils/misc.py:70-87).

The class histograms are exact integer counts (``torch.bincount`` instead of three ``torch.histc`` calls over float copies) kept
on the device; across ranks they are summed with one all-reduce of a (3, K) tensor instead of three.  AUPR / AUROC are computed on
the device too (sort by score + cumulative sums: the step-wise precision-recall sum and the trapezoidal ROC area with ties grouped,
i.e. what sklearn.metrics.average_precision_score / roc_auc_score compute); upstream hands the scores to sklearn on the host.
"""
import numpy as np
import torch


def intersection_and_union(output, target, k, ignore_index=-1):
    """-> (area_intersection, area_union, area_target), float32 (k,) on the inputs' device.  utils/misc.py:55-67; ``output`` is NOT
    modified (upstream overwrites ignored positions in place)."""
    output, target = output.reshape(-1), target.reshape(-1)
    valid = target != ignore_index
    o, t = output[valid].long(), target[valid].long()
    in_range = (o >= 0) & (o < k)
    area_output = torch.bincount(o[in_range], minlength=k)[:k]
    tr = (t >= 0) & (t < k)
    area_target = torch.bincount(t[tr], minlength=k)[:k]
    hit = (o == t) & in_range
    area_intersection = torch.bincount(o[hit], minlength=k)[:k]
    area_union = area_output + area_target - area_intersection
    return area_intersection.float(), area_union.float(), area_target.float()


def aupr_and_auroc(score, target, unknown_label, ignore_index=-1):
    """utils/misc.py:70-87 on the device: positives = points of the unknown classes, ignored points dropped; (None, None) when the
    batch holds no unknown point.  Returns python floats."""
    score, target = score.reshape(-1).double(), target.reshape(-1)
    valid = target != ignore_index
    score, target = score[valid], target[valid]
    pos = torch.isin(target, torch.as_tensor(list(unknown_label), device=target.device))
    n_pos = int(pos.sum())
    if n_pos == 0:
        return None, None
    n_neg = pos.numel() - n_pos
    order = torch.argsort(score, descending=True, stable=True)
    s, y = score[order], pos[order].double()
    tp, fp = torch.cumsum(y, 0), torch.cumsum(1.0 - y, 0)
    last = torch.ones_like(s, dtype=torch.bool)      # one operating point per DISTINCT score (ties share a threshold)
    last[:-1] = s[1:] != s[:-1]
    tp, fp = tp[last], fp[last]
    recall, precision = tp / n_pos, tp / (tp + fp)
    aupr = float(torch.sum(torch.diff(recall, prepend=recall.new_zeros(1)) * precision))
    if n_neg == 0:
        return aupr, float("nan")
    tpr, fpr = torch.cat([tp.new_zeros(1), tp / n_pos]), torch.cat([fp.new_zeros(1), fp / n_neg])
    auroc = float(torch.trapezoid(tpr, fpr))
    return aupr, auroc


class OpenSegEvaluator:
    """Accumulates what ``OpenSegEvaluator.eval`` logs (engines/hooks/evaluator.py:39-158): class histograms over the validation
    batches, mIoU / mAcc / allAcc over the KNOWN classes, mean AUPR / AUROC over the batches that contain unknown points."""

    def __init__(self, num_classes, unknown_label, ignore_index=-1):
        self.num_classes, self.unknown_label, self.ignore_index = num_classes, list(unknown_label), ignore_index
        self.mask_known = np.ones(num_classes, dtype=bool)
        self.mask_known[self.unknown_label] = False   # ~selected_mask(unknown_label, num_classes)
        self.reset()

    def reset(self):
        self.hist = None          # (3, K) float64 on the device: intersection | union | target
        self.aupr, self.auroc, self.losses = [], [], []

    @torch.no_grad()
    def update(self, seg_logits, score, segment_oracle, loss=None):
        """One validation batch: predictions = arg-max of the logits, ``segment_oracle`` = the labels incl. the unknown classes."""
        pred = seg_logits.max(1)[1]
        i, u, t = intersection_and_union(pred, segment_oracle, self.num_classes, self.ignore_index)
        h = torch.stack([i, u, t]).double()
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if multi:
            torch.distributed.all_reduce(h)
        self.hist = h if self.hist is None else self.hist + h
        pairs = [aupr_and_auroc(score, segment_oracle, self.unknown_label, self.ignore_index)]
        if multi:   # every rank's pair of this batch, None included, as `recognition_metric` gathers them (hooks/evaluator.py:199-221)
            gathered = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(gathered, pairs[0])
            pairs = gathered
        for a, r in pairs:
            if a is not None:
                self.aupr.append(a); self.auroc.append(r)
        if loss is not None:
            self.losses.append(float(loss))

    def summary(self):
        inter, union, target = (self.hist[j].cpu().numpy() for j in range(3))
        iou_class, acc_class = inter / (union + 1e-10), inter / (target + 1e-10)
        k = self.mask_known
        return dict(mIoU=float(np.mean(iou_class[k])), mAcc=float(np.mean(acc_class[k])),
                    allAcc=float(inter[k].sum() / (target[k].sum() + 1e-10)), iou_class=iou_class, acc_class=acc_class,
                    aupr=float(np.mean(self.aupr)) if self.aupr else float("nan"),
                    auroc=float(np.mean(self.auroc)) if self.auroc else float("nan"),
                    loss=float(np.mean(self.losses)) if self.losses else float("nan"))
