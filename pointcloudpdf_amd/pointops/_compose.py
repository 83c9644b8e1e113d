"""Query+group helpers and batch/offset converters of the reference's ``pointops`` package
(libs/pointops/functions/utils.py:5-121), built on ``_ops``."""
import torch

from ._ops import knn_query, ball_query, grouping


def knn_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, nsample=None, with_xyz=False):
    """utils.py:5-18 -> (grouped (m,nsample,[3+]c), idx)."""
    if idx is None:
        assert nsample is not None
        idx, _ = knn_query(nsample, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def ball_query_and_group(feat, xyz, offset=None, new_xyz=None, new_offset=None, idx=None, max_radio=None,
                         min_radio=0, nsample=None, with_xyz=False):
    """utils.py:21-41 (needs ball_query, which is off the hot path unless ``idx`` is supplied)."""
    if idx is None:
        assert nsample is not None and offset is not None
        assert max_radio is not None and min_radio is not None
        idx, _ = ball_query(nsample, max_radio, min_radio, xyz, offset, new_xyz, new_offset)
    return grouping(idx, feat, xyz, new_xyz, with_xyz), idx


def query_and_group(nsample, xyz, new_xyz, feat, idx, offset, new_offset, dilation=0, with_feat=True, with_xyz=True):
    """utils.py:44-99: dilated kNN grouping; relative xyz is NOT masked here (as upstream)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        total = 1 + (nsample - 1) * (dilation + 1)
        idx_all, _ = knn_query(total, xyz, offset, new_xyz, new_offset)
        ends, new_ends = offset.tolist(), new_offset.tolist()
        starts, new_starts = [0] + ends[:-1], [0] + new_ends[:-1]
        parts = []
        for b in range(offset.shape[0]):
            n_b = ends[b] - starts[b]
            soft = (n_b - 1) / (nsample - 1) - 1 if n_b < total else dilation
            cols = [int((soft + 1) * j) for j in range(nsample)]
            parts.append(idx_all[new_starts[b]: new_ends[b], cols])
        idx = torch.cat(parts, dim=0)
    if not with_feat:
        return idx
    m, c = new_xyz.shape[0], feat.shape[1]
    flat = idx.reshape(-1).long()
    grouped_xyz = xyz[flat, :].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    grouped_feat = feat[flat, :].view(m, nsample, c)
    if with_xyz:
        return torch.cat((grouped_xyz, grouped_feat), -1), idx
    return grouped_feat, idx


def offset2batch(offset):
    """utils.py:102-116: cumulative ends -> per-point scene id (int64, on offset's device)."""
    ends = [int(v) for v in offset.tolist()]
    sizes = [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]
    return torch.repeat_interleave(torch.arange(len(ends)), torch.tensor(sizes)).long().to(offset.device)


def batch2offset(batch):
    """utils.py:119-120"""
    return torch.cumsum(batch.bincount(), dim=0).int()
