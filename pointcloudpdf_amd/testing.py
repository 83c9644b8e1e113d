"""Test-time fragment voting (pointcept/engines/test.py:189-253).

The reference's tester splits a scene into GridSample "test" fragments (one point per voxel each), runs the segmentor and the
recognizer on every fragment, adds the softmax of the logits into a per-point vote, and averages the recognizer scores per point
with ``torch_scatter.scatter_mean`` (absent here and on the GPU box).  ``FragmentVoter`` keeps the same three accumulators on the
device and folds one fragment per call with a single kernel (``pdf_vote_accumulate``); ``result()`` returns what the tester
derives from them: ``pred = votes.argmax(1)``, ``score = sum / count`` (0 where a point was never visited -- scatter_mean's
convention)."""
import torch

from . import _native


class FragmentVoter:
    def __init__(self, num_points, num_classes, device):
        self.pred = torch.zeros(num_points, num_classes, dtype=torch.float32, device=device)   # test.py:207
        self.score_sum = torch.zeros(num_points, dtype=torch.float32, device=device)
        self.score_cnt = torch.zeros(num_points, dtype=torch.float32, device=device)

    @torch.no_grad()
    def add(self, seg_logits, index, score=None):
        """seg_logits (n, classes) of one fragment, index (n) its point ids in the full scene (distinct), score (n) or None."""
        be = _native.backend_for(seg_logits)
        be.vote_accumulate(seg_logits.float().contiguous(), None if score is None else score.float().contiguous(),
                           index.long().contiguous(), self.pred, self.score_sum, self.score_cnt)

    @torch.no_grad()
    def result(self):
        pred = self.pred.max(1)[1]                                            # test.py:242
        score = self.score_sum / self.score_cnt.clamp(min=1.0)               # scatter_mean (test.py:243-251)
        return pred, score


@torch.no_grad()
def fragment_inference(segmentor, recognizer_score_fn, data, fragments, num_classes):
    """Run ``segmentor`` (and ``recognizer_score_fn``) over GridSample test fragments of ONE scene and vote.

    data: dict with coord (N,3), feat (N,C) of the full scene on the device; fragments: list of (n_i,) index tensors
    (``voxelize.grid_sample(..., mode="test")["fragments"]``).  ``segmentor(input_dict)`` must return ``{"seg_logits": (n, K)}``
    (models/default.py:55-62); ``recognizer_score_fn(input_dict, seg_logits)`` returns per-point scores or None."""
    n = data["coord"].shape[0]
    voter = FragmentVoter(n, num_classes, data["coord"].device)
    for idx in fragments:
        part = dict(coord=data["coord"][idx].contiguous(), feat=data["feat"][idx].contiguous(),
                    offset=torch.tensor([idx.shape[0]], dtype=torch.int32, device=idx.device), offset_host=[int(idx.shape[0])])
        logits = segmentor(part)["seg_logits"]
        score = recognizer_score_fn(part, logits) if recognizer_score_fn is not None else None
        voter.add(logits, idx, score)
    return voter.result()
