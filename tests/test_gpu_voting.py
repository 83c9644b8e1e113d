"""GPU suite for SURVEY 8 row f-4: test-time fragment voting (pointcept/engines/test.py:189-253) -- the FragmentVoter kernel against
the tester's accumulation written out with plain torch ops, and the whole voxelise -> fragments -> eval -> vote path on a scene."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def reference_vote(n, k, parts):
    """engines/test.py:207-251 restated: pred[idx] += softmax(logits); score = scatter_mean(cat(score), cat(idx), dim_size=n)."""
    pred = torch.zeros(n, k, device="cuda")
    ssum = torch.zeros(n, device="cuda")
    scnt = torch.zeros(n, device="cuda")
    for logits, idx, score in parts:
        pred[idx, :] += torch.softmax(logits, -1)
        ssum.index_add_(0, idx, score)
        scnt.index_add_(0, idx, torch.ones_like(score))
    return pred.max(1)[1], ssum / scnt.clamp(min=1.0), pred


def test_fragment_voter_matches_tester_accumulation():
    from pointcloudpdf_amd.testing import FragmentVoter

    g = torch.Generator(device="cuda").manual_seed(4)
    n, k = 50000, 13
    parts = []
    for f in range(5):
        idx = torch.randperm(n, device="cuda", generator=g)[: 20000 + 1000 * f]      # distinct inside a fragment; ~18% never visited
        idx = idx[idx % 11 != 0]
        parts.append((torch.randn(idx.shape[0], k, device="cuda", generator=g) * 3, idx, torch.rand(idx.shape[0], device="cuda", generator=g)))
    voter = FragmentVoter(n, k, "cuda")
    for logits, idx, score in parts:
        voter.add(logits, idx, score)
    pred, score = voter.result()
    rp, rs, rvotes = reference_vote(n, k, parts)
    assert_close(voter.pred, rvotes, 1e-6, "votes")
    assert torch.equal(pred, rp)
    assert_close(score, rs, 1e-6, "score")
    assert (score[torch.arange(n, device="cuda") % 11 == 0] == 0).all()                # never-visited points: scatter_mean's zero


def test_voxelise_fragments_eval_vote_end_to_end():
    """A raw scene -> GridSample test fragments on the device -> PointTransformer-Seg26 + MSP score per fragment -> vote: every point
    gets a prediction, and a point's vote equals the sum of the softmax rows of the fragments that contain it."""
    from pointcloudpdf_amd import synthetic, testing, voxelize
    from pointcloudpdf_amd.registry import MODELS

    rng = np.random.default_rng(2)
    scene = synthetic.make_scene(6000, scene_id=77)
    raw = np.concatenate([scene["coord"], scene["coord"] + rng.normal(0, 0.004, scene["coord"].shape).astype(np.float32)])   # 2 points per voxel-ish
    color = np.concatenate([scene["color"], scene["color"]])
    coord = torch.from_numpy(raw).cuda()
    feat = torch.cat([coord, torch.from_numpy(color).cuda()], 1)
    off = torch.tensor([coord.shape[0]], dtype=torch.int32, device="cuda")
    frags = voxelize.grid_sample(coord, off, 0.08, mode="test")["fragments"]
    assert len(frags) >= 2
    seg = MODELS.build(dict(type="DefaultSegmentor", backbone=dict(type="PointTransformer-Seg26", in_channels=6, num_classes=13))).cuda().eval()
    synthetic.fill_parameters_deterministic(seg, seed=5)
    msp = lambda part, logits: -logits.log_softmax(-1).max(-1)[0]                       # MaxProbability "msp" score
    pred, score = testing.fragment_inference(seg, msp, dict(coord=coord, feat=feat), frags, 13)
    assert pred.shape == (coord.shape[0],) and score.shape == (coord.shape[0],)
    assert torch.isfinite(score).all() and (score > 0).all()                            # every point is in some fragment
    parts = []
    for idx in frags:
        part = dict(coord=coord[idx].contiguous(), feat=feat[idx].contiguous(), offset=torch.tensor([idx.shape[0]], dtype=torch.int32, device="cuda"))
        with torch.no_grad():
            lg = seg(part)["seg_logits"]
        parts.append((lg, idx, msp(part, lg)))
    rp, rs, _ = reference_vote(coord.shape[0], 13, parts)
    assert (pred == rp).float().mean() > 0.999                                          # (argmax flips only on exact vote ties)
    assert_close(score, rs, 1e-4, "score")
