"""The CSR-by-query attention ops of the reference's ``pointops2`` (libs/pointops2/functions/pointops.py) that
StratifiedTransformer's WindowAttention calls (stratified_transformer_v1m1_origin.py:277-341), on the C ABI of
include/pdfops.h.  Same names, positional orders, dtypes and autograd behaviour:

  attention_step1_v2(q, k, index1, index0_offsets, n_max) -> attn (M, h)                     pointops.py:170-258
  dot_prod_with_idx_v3(q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx)      pointops.py:632-755
  attention_step2_with_rel_pos_value_v2(attn, v, index0_offsets, n_max, index1, table, rel_idx) -> (N, h, d)   pointops.py:854-961

q / k / v are (N, h, d) fp32, indices int32, ``index0_offsets`` has N + 1 entries (edges of query i are
offsets[i] .. offsets[i+1]), tables are (L, h, d, 3), rel_idx is (M, 3).  Differences, all deliberate: any d (upstream throws
unless d is 16 or 32), outputs live on the inputs' device, kernels run on torch's current stream, argument errors raise.
The geometric ops under pointops2's own names (the model also calls these) are thin adapters over ``pointcloudpdf_amd.pointops``.
"""
import torch
from torch.autograd import Function

from .. import _native
from .. import pointops as _p1

_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def _be(t):
    return _native.backend_for(t)


class AttentionStep1_v2(Function):
    """libs/pointops2/functions/pointops.py:170-255"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, index1, index0_offsets, n_max):
        assert q.is_contiguous() and k.is_contiguous() and index0_offsets.is_contiguous() and index1.is_contiguous()
        assert n_max <= 1024
        out = _be(q).attention_step1_v2(q, k, index1, index0_offsets, n_max)
        ctx.n_max = int(n_max)
        ctx.save_for_backward(q, k, index0_offsets, index1)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        q, k, index0_offsets, index1 = ctx.saved_tensors
        gq, gk = _be(q).attention_step1_v2_backward(grad_output.contiguous(), q, k, index1, index0_offsets, ctx.n_max)
        return gq, gk, None, None, None


class DotProdWithIdx_v3(Function):
    """libs/pointops2/functions/pointops.py:632-752"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx):
        for t in (q, index_q_offsets, k, index_k, table_q, table_k, rel_idx):
            assert t.is_contiguous()
        assert table_k.shape[0] == table_q.shape[0]
        out = _be(q).dot_prod_with_idx_v3(q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx)
        ctx.n_max = int(n_max)
        ctx.save_for_backward(q, index_q_offsets, k, index_k, table_q, table_k, rel_idx)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        q, index_q_offsets, k, index_k, table_q, table_k, rel_idx = ctx.saved_tensors
        gq, gk, gtq, gtk = _be(q).dot_prod_with_idx_v3_backward(grad_output.contiguous(), q, index_q_offsets, ctx.n_max, k, index_k,
                                                                 table_q, table_k, rel_idx)
        return gq, None, None, gk, None, gtq, gtk, None


class AttentionStep2WithRelPosValue_v2(Function):
    """libs/pointops2/functions/pointops.py:854-958"""

    @staticmethod
    @_amp_fwd
    def forward(ctx, attn, v, index0_offsets, n_max, index1, table, rel_idx):
        for t in (attn, v, index0_offsets, index1, table, rel_idx):
            assert t.is_contiguous()
        out = _be(v).attention_step2_with_rel_pos_value_v2(attn, v, index0_offsets, n_max, index1, table, rel_idx)
        ctx.n_max = int(n_max)
        ctx.save_for_backward(attn, v, index0_offsets, index1, table, rel_idx)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        attn, v, index0_offsets, index1, table, rel_idx = ctx.saved_tensors
        ga, gv, gt = _be(v).attention_step2_with_rel_pos_value_v2_backward(grad_output.contiguous(), attn, v, index0_offsets, ctx.n_max,
                                                                            index1, table, rel_idx)
        return ga, gv, None, None, None, gt, None


class WindowLogits(Function):
    """attention_step1_v2(q, k, ...) + dot_prod_with_idx_v3(q, ..., k, ..., table_q, table_k, rel_idx) -- the sum WindowAttention.forward
    feeds the softmax (stratified_transformer_v1m1_origin.py:300-321) -- as one op: one pass over the key rows forward, and the two ops'
    gradients as single segmented passes backward (csrc/window_attention_bwd.hip).  HIP only (d = 16, L <= 64); ``window_logits`` falls back
    to the two reference ops elsewhere."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, index1, index0_offsets, table_q, table_k, rel_idx):
        for t in (q, k, index1, index0_offsets, table_q, table_k, rel_idx):
            assert t.is_contiguous()
        out = _be(q).window_logits(q, k, index1, index0_offsets, table_q, table_k, rel_idx)
        ctx.save_for_backward(q, k, index1, index0_offsets, table_q, table_k, rel_idx)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        q, k, index1, index0_offsets, table_q, table_k, rel_idx = ctx.saved_tensors
        gq, gk, gtq, gtk = _be(q).window_logits_backward(g.contiguous(), q, k, index1, index0_offsets, table_q, table_k, rel_idx)
        return gq, gk, None, None, gtq, gtk, None


def window_logits(q, k, index1, index0_offsets, n_max, table_q, table_k, rel_idx):
    be = _be(q)
    if getattr(be, "window_logits_supported", None) is not None and index1.shape[0] > 0 and be.window_logits_supported(q, k, table_q):
        return WindowLogits.apply(q, k, index1, index0_offsets, table_q, table_k, rel_idx)
    return (AttentionStep1_v2.apply(q, k, index1, index0_offsets, n_max)
            + DotProdWithIdx_v3.apply(q, index0_offsets, n_max, k, index1, table_q, table_k, rel_idx))


class WindowAttentionCore(Function):
    """Everything between the qkv Linear and the output projection of WindowAttention.forward (stratified_transformer_v1m1_origin.py:
    296-341: query scale, attention_step1_v2 + dot_prod_with_idx_v3, scatter_softmax, attention_step2_with_rel_pos_value_v2) as one
    autograd node on the (N, 3 C) output of the qkv Linear: q / k / v are column slices (no permute copy), the backward writes their
    gradients into the slices of one (N, 3 C) buffer (no per-slice zero-fill + add).  5 launches forward / 15 backward instead of
    ~9 / ~35, and only ``attn`` (M, h) is saved besides the inputs.  HIP only; ``window_attention_core`` composes the reference ops elsewhere."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, qkv, index1, index0_offsets, table_q, table_k, table_v, rel_idx, scale):
        out, attn = _be(qkv).window_attention_core(qkv, index1, index0_offsets, table_q, table_k, table_v, rel_idx, scale)
        ctx.save_for_backward(qkv, attn, index1, index0_offsets, table_q, table_k, table_v, rel_idx)
        ctx.scale = float(scale)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, go):
        qkv, attn, index1, index0_offsets, table_q, table_k, table_v, rel_idx = ctx.saved_tensors
        gqkv, gtq, gtk, gtv = _be(qkv).window_attention_core_backward(go.contiguous(), qkv, attn, index1, index0_offsets, table_q, table_k,
                                                                      table_v, rel_idx, ctx.scale)
        return gqkv, None, None, gtq, gtk, gtv, None, None


def window_attention_core(qkv, index1, index0_offsets, n_max, table_q, table_k, table_v, rel_idx, scale):
    """qkv (N, 3 C) -> (N, C): softmax_over_edges(<s q, k + T_q> + <k, T_k>) applied to (v + T_v); see WindowAttentionCore."""
    be = _be(qkv)
    if (getattr(be, "window_attention_core_supported", None) is not None and index1.shape[0] > 0
            and be.window_attention_core_supported(qkv, table_q, table_k, table_v)):
        return WindowAttentionCore.apply(qkv, index1, index0_offsets, table_q, table_k, table_v, rel_idx, float(scale))
    n = qkv.shape[0]
    L, h, d, _ = table_q.shape
    q, k, v = (t.contiguous() for t in qkv.reshape(n, 3, h, d).unbind(1))
    q = q * scale
    attn = segment_softmax(window_logits(q, k, index1, index0_offsets, n_max, table_q, table_k, rel_idx), index0_offsets)
    return AttentionStep2WithRelPosValue_v2.apply(attn, v, index0_offsets, n_max, index1, table_v, rel_idx).reshape(n, h * d)


attention_step1_v2 = AttentionStep1_v2.apply
dot_prod_with_idx_v3 = DotProdWithIdx_v3.apply
attention_step2_with_rel_pos_value_v2 = AttentionStep2WithRelPosValue_v2.apply


# ---- the geometric ops under pointops2's names / argument orders (libs/pointops2/functions/pointops.py:16-56, 964-1001, 1113-1127)
def furthestsampling(xyz, offset, new_offset):
    """pointops.py:16-34 -> idx (m) int32"""
    return _p1.farthest_point_sampling(xyz, offset, new_offset)


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.py:37-56: (nsample, xyz, new_xyz, offset, new_offset) -> (idx (m, nsample), dist (m, nsample) = sqrt(d2))"""
    if new_xyz is None:
        new_xyz, new_offset = xyz, offset
    return _p1.knn_query(nsample, xyz, offset, new_xyz, new_offset)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pointops.py:1113-1127 (same arithmetic as libs/pointops interpolation)"""
    return _p1.interpolation(xyz, new_xyz, feat, offset, new_offset, k)


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True, return_indx=False):
    """pointops.py:964-1001: kNN grouping, output (m, nsample, [3+]c) with the RELATIVE coordinates in front.  One fused
    gather (``pointops.grouping``) instead of the reference's index / subtract / cat chain; the only difference is in scenes
    with fewer than nsample points, where upstream's ``xyz[idx]`` wraps the -1 placeholder to the LAST row of the batch and
    this op gathers a zero row."""
    assert xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    assert new_xyz.is_contiguous()
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    out = _p1.grouping(idx, feat, xyz, new_xyz, with_xyz=use_xyz)
    return (out, idx) if return_indx else out


# ---- the v1 edge-list forms (arbitrary edge order, explicit index0) -- libs/pointops2/functions/pointops.py:93-167, 261-404,
# 407-629, 758-851.  Upstream runs them as edge-parallel kernels with one global atomic per (edge, channel); here the edge list is
# brought into CSR-by-query order once (stable sort of index0) and the v2 / v3 kernels above do the work; results return in the
# caller's edge order.  Autograd flows through the permutation gathers.
def _csr_by_query(index0, n_queries):
    i0 = index0.long()
    order = torch.sort(i0, stable=True)[1]
    counts = torch.bincount(i0, minlength=n_queries)
    offsets = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).int()
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(order.shape[0], device=order.device)
    return order, inverse, offsets


def _num_queries(index0, n_rows):
    """upstream sizes the output as index0.max() + 1 (a host sync); the CSR kernels want one offset per row of q / v."""
    nq = int(index0.max().item()) + 1 if index0.numel() else 0
    if nq > n_rows:
        raise ValueError(f"index0 refers to query {nq - 1} but there are only {n_rows} rows")
    return nq


def attention_step1(q, k, index0, index1):
    """pointops.py:93-167 -> attn (M, h) = <q[index0[m]], k[index1[m]]> per head"""
    order, inverse, offsets = _csr_by_query(index0, q.shape[0])
    return attention_step1_v2(q, k, index1[order].contiguous(), offsets, 0)[inverse]


def dot_prod_with_idx_v2(q, index_q, k, index_k, table_q, table_k, rel_idx):
    """pointops.py:476-629 -> (M, h)"""
    order, inverse, offsets = _csr_by_query(index_q, q.shape[0])
    out = dot_prod_with_idx_v3(q, offsets, 0, k, index_k[order].contiguous(), table_q, table_k, rel_idx[order].contiguous())
    return out[inverse]


def dot_prod_with_idx(q, index, table, rel_idx):
    """pointops.py:407-473 -> (M, h) = <q[index[m]], T(m)>: the two-sided op with a zero key table"""
    return dot_prod_with_idx_v2(q, index, q.detach(), index, table, torch.zeros_like(table), rel_idx)


def attention_step2_with_rel_pos_value(attn, v, index0, index1, table, rel_idx):
    """pointops.py:758-851 -> (index0.max() + 1, h, d)"""
    nq = _num_queries(index0, v.shape[0])
    order, _, offsets = _csr_by_query(index0, v.shape[0])
    out = attention_step2_with_rel_pos_value_v2(attn[order].contiguous(), v, offsets, 0, index1[order].contiguous(), table,
                                                rel_idx[order].contiguous())
    return out[:nq]


def attention_step2(attn, v, index0, index1):
    """pointops.py:261-335 -> (index0.max() + 1, h, d) = sum over the query's edges of attn * v[index1]"""
    h, d = v.shape[1], v.shape[2]
    table = v.new_zeros(1, h, d, 3)
    rel_idx = torch.zeros(index0.shape[0], 3, dtype=torch.int32, device=v.device)
    return attention_step2_with_rel_pos_value(attn, v, index0, index1, table, rel_idx)


attention_step2_v2 = attention_step2   # pointops.py:338-404: same semantics, different upstream launch shape


# ---- softmax over the edges of a query (torch_scatter.scatter_softmax(src, index_0, dim=0) for a CSR-ordered edge list) ----
class SegmentSoftmax(Function):
    """stratified_transformer_v1m1_origin.py:322-324; torch_scatter is an unvendored dependency of the reference, absent here."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, src, index0_offsets):
        y = _be(src).segment_softmax(src.contiguous(), index0_offsets.contiguous())
        ctx.save_for_backward(y, index0_offsets)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        y, offsets = ctx.saved_tensors
        return _be(y).segment_softmax_backward(y, grad_output.contiguous(), offsets), None


segment_softmax = SegmentSoftmax.apply
