"""Host side of the drop-in boundary: the public functions of the reference's ``pointops`` package
(libs/pointops/functions/__init__.py:1-14) re-authored on top of the C ABI in include/pdfops.h.

Same names, positional orders, dtypes and placeholders as upstream:
  * indices are int32, ``-1`` marks "scene has fewer than nsample points", ``offset`` holds cumulative ends;
  * ``knn_query`` returns ``(idx, sqrt(dist2))`` and is not differentiable (query.py:7-24);
  * ``grouping2 / interpolation2 / subtraction / aggregation / attention_*`` are autograd Functions whose
    backward returns exactly what the reference's does (incl. ``None`` for ``weight`` of the relation step,
    attention.py:62).
Differences, all deliberate: outputs are allocated on the inputs' device (no ``torch.cuda.*Tensor``), kernels run
on torch's current stream, errors are raised (the reference has none), and coordinate tensors tagged by a
``Geometry`` are served from its memo table instead of re-running kNN.
"""
import torch

# custom autograd nodes run in fp32 under autocast (the kernels are fp32; upstream's python ops promote to fp32 the same way)
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")
from torch.autograd import Function

from .. import _native
from ..geometry import tag_of, interpolation_weights


def _be(t):
    return _native.backend_for(t)


def _i32(t):
    return t if t.dtype == torch.int32 else t.int()


# ----------------------------------------------------------------------------- queries
class KNNQuery(Function):
    """libs/pointops/functions/query.py:7-24"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, nsample, xyz, offset, new_xyz=None, new_offset=None):
        if new_xyz is None or new_offset is None:
            new_xyz, new_offset = xyz, offset
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        ts, tq = tag_of(xyz), tag_of(new_xyz)
        if ts is not None and tq is not None and ts[0] is tq[0]:
            idx, dist = ts[0].knn_dist(nsample, ts[1], tq[1])
        else:
            idx, dist2 = _be(xyz).knn_query(nsample, xyz, new_xyz, _i32(offset).contiguous(), _i32(new_offset).contiguous())
            dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(idx, dist)
        return idx, dist


knn_query = KNNQuery.apply


class BallQuery(Function):
    """libs/pointops/functions/query.py:78-115 (argument order: nsample, max_radius, min_radius, ...)"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
        if new_xyz is None or new_offset is None:
            new_xyz, new_offset = xyz, offset
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        assert min_radius < max_radius
        idx, dist2 = _be(xyz).ball_query(nsample, max_radius, min_radius, xyz, new_xyz,
                                         _i32(offset).contiguous(), _i32(new_offset).contiguous())
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(idx, dist)
        return idx, dist


class RandomBallQuery(Function):
    """libs/pointops/functions/query.py:27-75: one ``torch.randperm`` per scene (drawn on ``offset``'s device from torch's
    global generator, as upstream), then the first nsample in-shell points along that permutation."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, nsample, max_radius, min_radius, xyz, offset, new_xyz=None, new_offset=None):
        if new_xyz is None or new_offset is None:
            new_xyz, new_offset = xyz, offset
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        assert min_radius < max_radius
        ends = [int(v) for v in offset.tolist()]
        order, start = [], 0
        for e in ends:
            order.append(torch.randperm(e - start, dtype=torch.int32, device=offset.device) + start)
            start = e
        order = torch.cat(order, dim=0).to(xyz.device)
        idx, dist2 = _be(xyz).ball_query(nsample, max_radius, min_radius, xyz, new_xyz,
                                         _i32(offset).contiguous(), _i32(new_offset).contiguous(), order=order.contiguous())
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(idx, dist)
        return idx, dist


ball_query = BallQuery.apply
random_ball_query = RandomBallQuery.apply


# ----------------------------------------------------------------------------- sampling
class FarthestPointSampling(Function):
    """libs/pointops/functions/sampling.py:7-24"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, xyz, offset, new_offset, offset_host=None, new_offset_host=None):
        assert xyz.is_contiguous()
        if offset_host is None:  # reference behaviour: host syncs (sampling.py:15-18)
            offset_host = [int(v) for v in offset.detach().cpu().tolist()]
        if new_offset_host is None:
            new_offset_host = [int(v) for v in new_offset.detach().cpu().tolist()]
        n_max, prev = 0, 0
        for e in offset_host:
            n_max, prev = max(n_max, e - prev), e
        idx = _be(xyz).farthest_point_sampling(
            xyz, _i32(offset).contiguous(), _i32(new_offset).contiguous(), n_max, new_offset_host[-1]
        )
        ctx.mark_non_differentiable(idx)
        return idx


def farthest_point_sampling(xyz, offset, new_offset):
    return FarthestPointSampling.apply(xyz, offset, new_offset)


# ----------------------------------------------------------------------------- grouping2
class Grouping(Function):
    """libs/pointops/functions/grouping.py:7-33"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, input, idx):
        assert input.is_contiguous() and idx.is_contiguous()
        ctx.n = input.shape[0]
        ctx.save_for_backward(idx)
        return _be(input).grouping_forward(input, idx)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        return _be(grad_output).grouping_backward(grad_output.contiguous(), idx, ctx.n), None


grouping2 = Grouping.apply


class _GroupFused(Function):
    """Fused twin of the python grouping() (grouping.py:36-60): gather + relative xyz + mask + cat in one pass."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, feat, xyz, new_xyz, idx, with_xyz):
        ctx.shape = (feat.shape[0], feat.shape[1], bool(with_xyz))
        ctx.save_for_backward(idx)
        return _be(feat).group_forward(feat, xyz, new_xyz, idx, bool(with_xyz))

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        n, c, with_xyz = ctx.shape
        return _be(grad_output).group_backward(grad_output.contiguous(), idx, n, c, with_xyz), None, None, None, None


def grouping(idx, feat, xyz, new_xyz=None, with_xyz=False):
    """libs/pointops/functions/grouping.py:36-60 -> (m, nsample, c) or (m, nsample, 3+c); idx -1 gathers zeros."""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and feat.is_contiguous()
    be = _be(feat)
    if hasattr(be, "group_forward") and feat.dtype == torch.float32:  # (fp64 inputs: torch composition, used by accuracy probes)
        if with_xyz:
            assert new_xyz.is_contiguous()
        return _GroupFused.apply(feat, xyz, new_xyz, idx.contiguous(), with_xyz)
    # composition as upstream (used by backends without the fused entry point)
    m, nsample, c = idx.shape[0], idx.shape[1], feat.shape[1]
    xyz_p = torch.cat([xyz, xyz.new_zeros(1, 3)], dim=0)
    feat_p = torch.cat([feat, feat.new_zeros(1, c)], dim=0)
    flat = idx.reshape(-1).long()
    grouped_feat = feat_p[flat, :].view(m, nsample, c)
    if not with_xyz:
        return grouped_feat
    mask = torch.sign(idx + 1)
    grouped_xyz = xyz_p[flat, :].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    grouped_xyz = grouped_xyz * mask.unsqueeze(-1).to(grouped_xyz.dtype)
    return torch.cat((grouped_xyz, grouped_feat), -1)


# ----------------------------------------------------------------------------- interpolation
class _InterpolateIdx(Function):
    """out = sum_k feat[idx[:,k]] * weight[:,k] with a scatter backward (interpolation.py:20-21 / :25-59)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, input, idx, weight):
        ctx.m = input.shape[0]
        ctx.save_for_backward(idx, weight)
        return _be(input).interpolation_forward(input, idx, weight)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        return _be(grad_output).interpolation_backward(grad_output.contiguous(), idx, weight, ctx.m), None, None


def _interp_tables(xyz, new_xyz, offset, new_offset, k):
    tc, tf = tag_of(xyz), tag_of(new_xyz)
    if tc is not None and tf is not None and tc[0] is tf[0]:
        return tc[0].interp(tc[1], tf[1], k)
    idx, dist2 = _be(xyz).knn_query(k, xyz, new_xyz, _i32(offset).contiguous(), _i32(new_offset).contiguous())
    return idx, interpolation_weights(dist2)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """libs/pointops/functions/interpolation.py:8-22: coords (m,3) -> new_xyz (n,3), feat (m,c) -> (n,c)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, weight = _interp_tables(xyz, new_xyz, offset, new_offset, k)
    if feat.dtype == torch.float64:  # accuracy probes (fp64 yardstick runs): upstream's own formulation (:18-21), zero row for idx -1
        feat_p = torch.cat([feat, feat.new_zeros(1, feat.shape[1])], dim=0)
        out = feat.new_zeros(new_xyz.shape[0], feat.shape[1])
        for i in range(k):
            out = out + feat_p[idx[:, i].long(), :] * weight[:, i].double().unsqueeze(-1)
        return out
    if feat.dtype != torch.float32:  # autocast: upstream accumulates feat[idx] * weight (fp32) into an fp32 tensor
        feat = feat.float()
    return _InterpolateIdx.apply(feat, idx, weight)


class Interpolation(Function):
    """libs/pointops/functions/interpolation.py:25-59"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        idx, weight = _interp_tables(xyz, new_xyz, offset, new_offset, k)
        ctx.m = input.shape[0]
        ctx.save_for_backward(idx, weight)
        return _be(input).interpolation_forward(input, idx, weight)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        gi = _be(grad_output).interpolation_backward(grad_output.contiguous(), idx, weight, ctx.m)
        return None, None, gi, None, None, None


interpolation2 = Interpolation.apply


# ----------------------------------------------------------------------------- subtraction / aggregation
class Subtraction(Function):
    """libs/pointops/functions/subtraction.py:7-38"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, input1, input2, idx):
        assert input1.is_contiguous() and input2.is_contiguous()
        ctx.n2 = input2.shape[0]
        ctx.save_for_backward(idx)
        return _be(input1).subtraction_forward(input1, input2, idx.contiguous())

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        (idx,) = ctx.saved_tensors
        g1, g2 = _be(grad_output).subtraction_backward(idx, grad_output.contiguous(), ctx.n2)
        return g1, g2, None


subtraction = Subtraction.apply


class Aggregation(Function):
    """libs/pointops/functions/aggregation.py:7-57"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, input, position, weight, idx):
        assert input.is_contiguous() and position.is_contiguous() and weight.is_contiguous()
        ctx.save_for_backward(input, position, weight, idx)
        return _be(input).aggregation_forward(input, position, weight, idx.contiguous())

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        input, position, weight, idx = ctx.saved_tensors
        gi, gp, gw = _be(grad_output).aggregation_backward(input, position, weight, idx, grad_output.contiguous())
        return gi, gp, gw, None


aggregation = Aggregation.apply


# ----------------------------------------------------------------------------- attention steps
class AttentionRelationStep(Function):
    """libs/pointops/functions/attention.py:12-62"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, query, key, weight, index_target, index_refer):
        assert query.is_contiguous() and key.is_contiguous() and weight.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        it, ir = _i32(index_target), _i32(index_refer)
        ctx.save_for_backward(query, key, weight, it, ir)
        return _be(query).attention_relation_step_forward(query, key, weight, it, ir)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        query, key, weight, it, ir = ctx.saved_tensors
        gq, gk, _ = _be(query).attention_relation_step_backward(query, key, weight, it, ir, grad_output.contiguous())
        return gq, gk, None, None, None  # grad_weight is computed but dropped upstream too (attention.py:62)


class AttentionFusionStep(Function):
    """libs/pointops/functions/attention.py:65-116"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, weight, value, index_target, index_refer):
        assert weight.is_contiguous() and value.is_contiguous()
        assert index_target.is_contiguous() and index_refer.is_contiguous()
        assert index_target.shape[0] == index_refer.shape[0]
        it, ir = _i32(index_target), _i32(index_refer)
        ctx.save_for_backward(weight, value, it, ir)
        return _be(value).attention_fusion_step_forward(weight, value, it, ir)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        weight, value, it, ir = ctx.saved_tensors
        gw, gv = _be(value).attention_fusion_step_backward(weight, value, it, ir, grad_output.contiguous())
        return gw, gv, None, None


attention_relation_step = AttentionRelationStep.apply
attention_fusion_step = AttentionFusionStep.apply
