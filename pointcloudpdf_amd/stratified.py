"""StratifiedTransformer (``ST-v1m1``) and its PDF U-decoder (``ST-v1m1-Recognizer``) on the window-attention kernels of
``pointcloudpdf_amd.pointops2`` -- SURVEY.md 8 row f-1 / BASELINE config 5.

Mirrors pointcept/models/stratified_transformer/stratified_transformer_v1m1_origin.py and
pointcept/recognizers/recognizer_model/st_v1m1.py class by class (same constructor arguments, same submodule / parameter names:
``stem_layer.{i}.kpconv.weight``, ``layers.{i}.blocks.{j}.attn.{qkv,proj,relative_pos_*_table}``, ``layers.{i}.downsample.{norm,linear}``,
``upsamples.{i}.linear{1,2}``, ``classifier``; hook names ``backbone.upsamples.{i}`` with ``forward_input`` / ``forward_output``), so
reference checkpoints load and the reference's hook configuration resolves.

What runs where: window attention = ``pointops2`` HIP kernels (attention_step1_v2, dot_prod_with_idx_v3, CSR segment softmax,
attention_step2_with_rel_pos_value_v2); FPS / kNN / grouping / interpolation = the ``pointops`` HIP kernels; the window partition
(upstream ``grid_sample`` / ``get_indice_pairs`` + the sort of the edges by query: unique / argsort / dense boolean-mask expansion) is
built per QUERY by csrc/window_edges.hip from per-point window keys (``window_keys``), with one host read per partition (edge count and
longest row); the reference construction itself lives in oracle/window_tables.py, where the tests compare the two bit for bit.

Third-party pieces the reference imports and this image lacks -- all unvendored, unversioned, hence **parity unpinned** -- are restated
from their documented behaviour in this file: ``torch_scatter.scatter_softmax`` (-> the CSR segment-softmax kernel),
``torch_geometric.nn.pool.voxel_grid`` (``_voxel_grid``), ``timm`` ``DropPath`` / ``trunc_normal_``, ``torch_points_kernels.ball_query``
(-> ``pseudo_label.radius_neighbors``: first ``max_neighbor`` points in index order within the radius, -1 padded) and the
``torch_points3d`` ``KPConvLayer`` / ``FastBatchNorm1d`` of the stem (rigid kernel points with linear influence; the kernel-point
disposition of torch_points3d is the result of an offline optimisation shipped as a data file, so a closed-form layout -- centre +
Fibonacci sphere -- stands in).  The reference's OWN code around them is pinned by tests/golden/model_stratified_*.npz (the
reference classes imported and run with these same stand-ins injected).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native, dense
from .pointops import _ops as _p1_ops
from .pointops2 import pointops
from .registry import MODELS


# PDFOPS_ST_LINEAR=hip routes qkv / proj / fc1 / fc2 through csrc/rowlin.hip.  Measured in round 4 (2 x 80k points, one process): 73.6 ms per
# step against 67.6 with the library GEMMs + split-K weight gradient -- the widths of this model (48 / 96 / 192 / 384, 3x and 4x expansions)
# only have the generic tiled kernel there, not the rl2:: kernels of PT-v1's 32 .. 512 square layers.  Default: library GEMMs.
ST_HIP_LINEAR = os.environ.get("PDFOPS_ST_LINEAR", "torch") == "hip"


class _Linear(nn.Linear):
    """nn.Linear (same parameters, same ``state_dict`` keys) whose WEIGHT gradient over many rows runs as a batched split-K product
    (dense._LinearSplitK): the library's dW = dY^T X with 10^5..10^6 rows is one 32x32 macro-tile on a single workgroup (0.5 ms per
    call at 160k x 48 -> 144; 7.8 ms of the 93 ms ST-v1m1 step, profiles/r03_i_stratified_kernel_trace_stats.txt)."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() in (2, 3) and dense.HIP_LINEAR and ST_HIP_LINEAR:
            # (round 4) forward, input gradient and weight gradient on the library's own matrix-core kernels (csrc/rowlin.hip): no rocBLAS
            # GEMM left under qkv / proj / fc1 / fc2 and the heads
            y = dense.linear(self, x.view(-1, x.shape[-1]))
            return y if x.dim() == 2 else y.view(x.shape[0], x.shape[1], -1)
        if x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.is_contiguous() and x.dim() in (2, 3):
            rows = x.shape[0] if x.dim() == 2 else x.shape[0] * x.shape[1]
            if rows >= (1024 if dense._OWN_WGRAD else dense._MIN_ROWS):   # (the own weight-gradient kernel pays from ~1k rows on)
                y = dense._LinearSplitK.apply(x.view(rows, x.shape[-1]), self.weight, self.bias)
                return y if x.dim() == 2 else y.view(x.shape[0], x.shape[1], -1)
        return F.linear(x, self.weight, self.bias)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, quant_size, rel_query=True, rel_key=False, rel_value=False, qkv_bias=True,
                 qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.window_size, self.quant_size = window_size, quant_size
        self.rel_query, self.rel_key, self.rel_value = rel_query, rel_key, rel_value
        quant_grid_length = int((2 * window_size + 1e-4) // quant_size)                  # :216
        for flag, name in ((rel_query, "relative_pos_query_table"), (rel_key, "relative_pos_key_table"),
                           (rel_value, "relative_pos_value_table")):
            if flag:
                table = nn.Parameter(torch.zeros(2 * quant_grid_length, num_heads, head_dim, 3))
                nn.init.trunc_normal_(table, std=0.02)
                setattr(self, name, table)
        self.quant_grid_length = quant_grid_length
        self.qkv = _Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop, inplace=True)
        self.proj = _Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop, inplace=True)

    def relative_position_index(self, xyz, index_0, index_1):
        """:282-292 -- quantised offset of every edge, (M, 3) in [0, 2 * quant_grid_length)"""
        rel = xyz[index_0.long()] - xyz[index_1.long()]
        # Divisors as 0-dim DEVICE tensors: torch divides by a python scalar on the GPU by multiplying with its reciprocal, which
        # quantises ~4e-5 of the edges into the neighbouring table row compared with the IEEE division of the CPU path (measured:
        # 17 of 443,598 entries at 5,500 points).  Upstream's own CPU and CUDA runs differ in exactly that way; the true division keeps
        # this op device-independent (and equal to the CPU-generated fixture).
        rel = torch.round(rel * 100000) / rel.new_tensor(100000.0)
        return torch.div(rel + 2 * self.window_size - 1e-4, rel.new_tensor(float(self.quant_size)), rounding_mode="trunc")

    def checked_relative_position_index(self, xyz, index_0, index_1):
        rel_idx = self.relative_position_index(xyz, index_0, index_1)
        assert (rel_idx >= 0).all() and (rel_idx <= 2 * self.quant_grid_length - 1).all()
        return rel_idx.int().contiguous()

    def forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max, rel_idx=None):
        """``rel_idx``: the (M, 3) int32 table rows of the edges when the caller already holds them (they depend on coordinates only:
        BasicLayer computes them once per window partition instead of once per block)."""
        n, c = feats.shape
        assert index_0.shape[0] == index_1.shape[0]
        i1, off = index_1.int().contiguous(), index_0_offsets.int().contiguous()
        if rel_idx is None:
            rel_idx = self.checked_relative_position_index(xyz, index_0, index_1)
        if self.rel_query and self.rel_key and self.rel_value:   # the configured model: everything up to the projection as one node
            x = pointops.window_attention_core(self.qkv(feats).float(), i1, off, n_max, self.relative_pos_query_table.float(),
                                               self.relative_pos_key_table.float(), self.relative_pos_value_table.float(), rel_idx, self.scale)
            return self.proj_drop(self.proj(x))
        qkv = self.qkv(feats).reshape(n, 3, self.num_heads, c // self.num_heads).permute(1, 0, 2, 3).contiguous()
        query, key, value = qkv[0], qkv[1], qkv[2]
        query = query * self.scale
        if self.rel_query and self.rel_key:   # attention_step1_v2 + dot_prod_with_idx_v3 (:300-321) as one op where the backend has it
            logits = pointops.window_logits(query.float(), key.float(), i1, off, n_max, self.relative_pos_query_table.float(),
                                            self.relative_pos_key_table.float(), rel_idx)
        else:
            logits = pointops.attention_step1_v2(query.float(), key.float(), i1, off, n_max)
            if self.rel_query:
                logits = logits + pointops.dot_prod_with_idx(query.float(), index_0.int(), self.relative_pos_query_table.float(), rel_idx)
            elif self.rel_key:
                logits = logits + pointops.dot_prod_with_idx(key.float(), index_1.int(), self.relative_pos_key_table.float(), rel_idx)
        attn = pointops.segment_softmax(logits, off)                                       # scatter_softmax(src, index_0, dim=0), :322-324
        # (upstream constructs ``attn_drop`` but never applies it -- :246 vs :322-341 -- so neither does this forward)
        if self.rel_value:
            x = pointops.attention_step2_with_rel_pos_value_v2(attn.float(), value.float(), off, n_max, i1,
                                                               self.relative_pos_value_table.float(), rel_idx)
        else:
            x = pointops.attention_step2(attn.float(), value.float(), index_0.int(), index_1.int())
        x = self.proj(x.view(n, c))
        return self.proj_drop(x)


# ------------------------------------------------------------------------------------------------------------------
# stand-ins for the absent third-party pieces (see the module docstring)
# ------------------------------------------------------------------------------------------------------------------
def _colminmax(x):
    """(min, max) over the rows of an (N, c) tensor with small c: from the transposed copy (c contiguous rows reduce in microseconds; the
    column reduction of an (N, 3) tensor takes torch 0.2-1.4 ms at N = 0.2-2.4 M).  Same values: min / max are exact."""
    return torch.aminmax(x.t().contiguous(), dim=1)


def _voxel_grid(pos, batch, size, start=None):
    """torch_geometric.nn.pool.voxel_grid -> torch_cluster.grid_cluster: the batch index is appended as a 4th coordinate of cell size
    1; cell = floor((pos - start) / size) per axis; cluster id = sum_d cell_d * prod_{e<d} (floor((end_e - start_e) / size_e) + 1)
    (x fastest).  ``start`` None -> the per-axis minimum, ``end`` = the per-axis maximum."""
    pos4 = torch.cat([pos, batch.unsqueeze(-1).to(pos.dtype)], dim=-1)
    size4 = torch.cat([size.to(pos.dtype), size.new_ones(1).to(pos.dtype)])
    lo4, end4 = _colminmax(pos4)
    start4 = lo4 if start is None else torch.cat([start.to(pos.dtype), pos.new_zeros(1)])
    cell = torch.div(pos4 - start4, size4, rounding_mode="floor").long()
    num = (torch.div(end4 - start4, size4, rounding_mode="floor").long() + 1).clamp_(min=1)
    stride = torch.cumprod(torch.cat([num.new_ones(1), num[:-1]]), 0)
    return (cell * stride).sum(-1)


class DropPath(nn.Module):
    """timm.models.layers.DropPath: stochastic depth per sample (row), scaled by 1 / keep_prob in training."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


class FastBatchNorm1d(nn.Module):
    """torch_points3d.core.common_modules.FastBatchNorm1d on (N, C) rows: a BatchNorm1d under the attribute ``batch_norm``."""

    def __init__(self, num_features, momentum=0.1, **kw):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(num_features, momentum=momentum, **kw)

    def forward(self, x):
        return self.batch_norm(x)


def _kernel_points(radius, n_points=15):
    """Centre + (n_points - 1) points of a Fibonacci sphere at 0.66 * radius.  Stands in for torch_points3d's ``load_kernels``
    (an optimised disposition read from a data file; parity unpinned)."""
    import math

    pts = [[0.0, 0.0, 0.0]]
    m = n_points - 1
    golden = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(m):
        z = 1.0 - 2.0 * (i + 0.5) / m
        r = math.sqrt(max(0.0, 1.0 - z * z))
        pts.append([r * math.cos(golden * i), r * math.sin(golden * i), z])
    k = torch.tensor(pts, dtype=torch.float32)
    k[1:] *= 0.66 * radius
    return k


class KPConvLayer(nn.Module):
    """Rigid KPConv (torch_points3d.modules.KPConv.kernels.KPConvLayer, KPConv_ops): 15 kernel points, linear influence
    ``max(0, 1 - |y - x_k| / KP_extent)``, "sum" aggregation; parameter ``weight`` (K, C_in, C_out), buffer ``K_points`` (K, 3)."""

    _INFLUENCE_TO_RADIUS = 1.5

    def __init__(self, num_inputs, num_outputs, point_influence, n_kernel_points=15, add_one=False, **kw):
        super().__init__()
        self.kernel_radius = self._INFLUENCE_TO_RADIUS * point_influence
        self.KP_extent = point_influence
        self.add_one = add_one
        self.num_inputs = num_inputs + int(add_one)
        self.num_outputs = num_outputs
        self.register_buffer("K_points", _kernel_points(self.kernel_radius, n_kernel_points))
        weight = torch.empty(n_kernel_points, self.num_inputs, num_outputs)
        nn.init.xavier_normal_(weight)
        self.weight = nn.Parameter(weight)

    FUSED = os.environ.get("PDFOPS_KPCONV", "hip") != "torch"   # torch: the composed form below on the device as well (A/B runs)

    def forward(self, query_points, support_points, neighbors, x):
        """neighbors (N, M) indices into support_points, -1 = no neighbour (a far "shadow" point with zero features)."""
        if self.add_one:
            x = torch.cat([x, x.new_ones(x.shape[0], 1)], 1)
        if (self.FUSED and x.is_cuda and x.dtype == torch.float32 and _native.hip_backend().kpconv_supported(self.K_points.shape[0], x.shape[1])):
            return _KPConvFn.apply(query_points.contiguous(), support_points.contiguous(), neighbors.to(torch.int32).contiguous(),
                                   x.contiguous(), self.K_points, self.weight, float(self.KP_extent))
        n_s = support_points.shape[0]
        nb = torch.where(neighbors < 0, torch.full_like(neighbors, n_s), neighbors).long()
        sp = torch.cat([support_points, support_points.new_full((1, 3), 1e6)], 0)
        xs = torch.cat([x, x.new_zeros(1, x.shape[1])], 0)
        diff = sp[nb] - query_points.unsqueeze(1)                                         # (N, M, 3)
        d2 = ((diff.unsqueeze(2) - self.K_points) ** 2).sum(-1)                           # (N, M, K)
        w = torch.clamp(1.0 - torch.sqrt(d2) / self.KP_extent, min=0.0).transpose(1, 2)   # (N, K, M)
        weighted = torch.matmul(w, xs[nb])                                                # (N, K, C_in)
        return torch.einsum("nkc,kco->no", weighted, self.weight)


class _KPConvFn(torch.autograd.Function):
    """KPConvLayer.forward on the device without the (N, M, K) intermediates: influence-weighted gather per (query, kernel point)
    (csrc/kpconv.hip) + one plain product with the (K * C_in, C_out) weight; the backward is the two adjoint products + the scatter."""

    @staticmethod
    @dense._amp_fwd
    def forward(ctx, query, support, neighbors, x, k_points, weight, extent):
        be = _native.hip_backend()
        kc, o = weight.shape[0] * weight.shape[1], weight.shape[2]
        weighted = be.kpconv_gather(query, support, neighbors, x, k_points, extent)                 # (N, K * C_in)
        out = be.rowlin(weighted, weight.detach().reshape(kc, o), transpose_w=True)[0]              # weighted @ W
        ctx.save_for_backward(query, support, neighbors, k_points, weight, weighted)
        ctx.cfg = (extent, x.shape[0], x.shape[1])
        return out

    @staticmethod
    @dense._amp_bwd
    def backward(ctx, g):
        query, support, neighbors, k_points, weight, weighted = ctx.saved_tensors
        extent, rows, cin = ctx.cfg
        be = _native.hip_backend()
        kc, o = weight.shape[0] * weight.shape[1], weight.shape[2]
        g = g.contiguous()
        wm = weight.detach().reshape(kc, o)
        gx = gw = None
        if ctx.needs_input_grad[3]:
            g_weighted = be.rowlin(g, wm)[0]                                                         # g @ W^T  (N, K * C_in)
            gx = be.kpconv_scatter(query, support, neighbors, g_weighted, k_points, extent, rows, cin)
        if ctx.needs_input_grad[5]:
            gw = be.rowlin_wgrad(weighted, g, None, False, False)[0].view_as(weight)                 # weighted^T @ g  (K * C_in, C_out)
        return None, None, None, gx, None, gw, None


def offset2batch(offset, n=None):
    """:27-42 (without the per-scene python lists).  ``n``: the number of points when the caller knows it (``coord.shape[0]``): without it
    ``repeat_interleave`` reads the total from the device -- the host waits for everything queued before it (6.7 ms per step at
    2 x 80k points, tools/st_host_profile.py)."""
    sizes = torch.diff(offset.long(), prepend=offset.new_zeros(1).long())
    return torch.repeat_interleave(torch.arange(offset.shape[0], device=offset.device), sizes, output_size=n)


# ------------------------------------------------------------------------------------------------------------------
# window partitions: keys per point here (elementwise), the edge tables by the backend (csrc/window_edges.hip)
# ------------------------------------------------------------------------------------------------------------------
def window_keys(xyz, batch, window_size, xyz_min, parity):
    """What one Swin block's partition is made of, per POINT (stratified_transformer_v1m1_origin.py:468-499 + :91-94): the voxel id of
    the fine window partition, of the coarse (2 x window) one -- torch_geometric's ``voxel_grid`` on the plain coordinates for even blocks,
    on the coordinates shifted by half a window from ``xyz_min`` for odd ones -- and the fine-window CELL of :91-94 (the predicate "lies
    in another fine window" compares those cells), packed into one integer.  -> (kf, kc, wk), int64 each."""
    new_window_size = 2 * window_size
    if parity % 2 == 0:
        kf, kc = _voxel_grid(xyz, batch, window_size, start=None), _voxel_grid(xyz, batch, new_window_size, start=None)
        shift = 0.0
    else:
        kf = _voxel_grid(xyz + 1 / 2 * window_size, batch, window_size, start=xyz_min)
        kc = _voxel_grid(xyz + 1 / 2 * new_window_size, batch, new_window_size, start=xyz_min)
        shift = 1 / 2 * window_size
    wc = torch.div(xyz - xyz_min + shift, window_size, rounding_mode="trunc").long()
    # (the cells are small non-negative integers: one packed key per point gives `(a != b).any(-1)` of :93-94 with one comparison)
    return kf, kc, (wc[:, 0] << 42) | (wc[:, 1] << 21) | wc[:, 2]


class Mlp(nn.Module):
    """:130-153"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = _Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = _Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop, inplace=True)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


def _strided_counts(ends, fn):
    out, count, prev = [], 0, 0
    for e in ends:
        count += fn(e - prev)
        prev = e
        out.append(count)
    return out


def _strided_offsets(offset, fn):
    return torch.tensor(_strided_counts([int(v) for v in offset.detach().cpu().tolist()], fn), dtype=torch.int32, device=offset.device)


def _fps_pair(xyz, off, ends, fn_keys, fn_down):
    """(key indices, their scene ends, down-sample indices, their scene ends) of one level from ONE farthest-point run.  Sample j of the
    reference's kernel is the arg-max after j - 1 insertions whatever the requested count (sampling_cuda_kernel.cu:42-127: the loop bound
    is the only use of m; the block size / tie rule depends on n alone), so the shorter subset is the per-scene PREFIX of the longer one:
    one chain of max(keys, down) dependent steps per scene instead of two (bit-identical indices, tests/test_gpu_pointops2.py)."""
    k_ends, d_ends = _strided_counts(ends, fn_keys), _strided_counts(ends, fn_down)
    k_cnt = [e - (k_ends[i - 1] if i else 0) for i, e in enumerate(k_ends)]
    d_cnt = [e - (d_ends[i - 1] if i else 0) for i, e in enumerate(d_ends)]
    l_cnt = [max(a, b) for a, b in zip(k_cnt, d_cnt)]
    l_ends = [sum(l_cnt[:i + 1]) for i in range(len(l_cnt))]
    l_off = torch.tensor(l_ends, dtype=torch.int32).to(xyz.device, non_blocking=True)
    long_idx = pointops.furthestsampling(xyz, off, l_off)

    def prefix(cnt):
        if cnt == l_cnt:
            return long_idx
        parts = [long_idx[(l_ends[i] - l_cnt[i]):(l_ends[i] - l_cnt[i]) + c] for i, c in enumerate(cnt)]
        return (parts[0] if len(parts) == 1 else torch.cat(parts)).contiguous()

    return prefix(k_cnt), k_ends, prefix(d_cnt), d_ends


class StratifiedGeometry:
    """The coordinate-only chain of one StratifiedTransformer forward: per level the FPS subset that supplies the window keys
    (n // downsample_scale + 1 points per scene, BasicLayer :479-489) and the TransitionDown sample (int(n * ratio) + 1 points, :165-174),
    whose points are the next level's coordinates.  None of it depends on features, so -- as for PointTransformer-V1's Geometry --
    it can run ahead of the step on a side stream (FPS is a serial chain per scene: 7 launches, ~60 ms of the 179 ms step at 2 x 80k
    points when it runs inline).  Results are bit-identical to the inline calls (same kernel, same inputs).  With ``offset_host`` (the
    scene ends as Python ints, as the collate function has them) no call of the chain waits for the device."""

    def __init__(self, coord, offset, offset_host=None, downsample_scale=8, ratio=0.25, num_layers=4, stem_transformer=True, k=16, up_k=3,
                 ball=None):
        self.coord, self.offset = coord.contiguous(), offset.int()
        self.offset_host = [int(v) for v in (offset_host if offset_host is not None else offset.detach().cpu().tolist())]
        self.cfg = (downsample_scale, ratio, num_layers, stem_transformer)
        self.nn_cfg = (k, up_k, ball)   # TransitionDown's k, Upsample's interpolation k, (radius, max_neighbor) of the KPConv stem
        self.samples = {}        # ("keys" | "down", level) -> (int32 indices into that level's points, int32 scene ends of the subset)
        self.windows = {}        # level -> BasicLayer.window_tables (filled by precompute(layers=...) only: data-dependent shapes)
        # The neighbour searches of the forward (filled together with the windows): ("ball",) -> the KPConv stem's radius table;
        # ("td", level) -> (coordinates of the level's TransitionDown sample = the next level's points, their k nearest points of the
        # level); ("up", level) -> (idx, weight) of the interpolation from level + 1 back onto level (Upsample :558-579 -- the backbone's
        # stack and the recognizer's walk the same pairs).  Coordinates only, static shapes.
        self.neighbors = {}

    def _level_neighbors(self, key, xyz, off, n_xyz, n_off):
        k, up_k, _ = self.nn_cfg
        self.neighbors[("td", key)] = (n_xyz, _native.backend_for(xyz).knn_query(k, xyz, n_xyz, off.int().contiguous(), n_off.int().contiguous())[0])
        self.neighbors[("up", key)] = _p1_ops._interp_tables(n_xyz, xyz, n_off, off, up_k)
        if xyz.is_cuda:   # the entries of both tables grouped by source row: what the backward's segmented sums walk (one stable sort
            # per table, cached on the idx tensor -- csrc/seg_gather.hip; built lazily inside the backward otherwise)
            _native.inverse_table(self.neighbors[("td", key)][1], xyz.shape[0])
            _native.inverse_table(self.neighbors[("up", key)][0], n_xyz.shape[0])

    def _ball(self):
        from .pseudo_label import radius_neighbors

        if self.nn_cfg[2] is not None:
            self.neighbors[("ball",)] = radius_neighbors(self.coord, self.offset, *self.nn_cfg[2])

    def _sample(self, key, xyz, off, ends, fn):
        n_ends = _strided_counts(ends, fn)
        n_off = torch.tensor(n_ends, dtype=torch.int32).to(xyz.device, non_blocking=True)
        self.samples[key] = (pointops.furthestsampling(xyz, off, n_off), n_off)
        return self.samples[key] + (n_ends,)

    def _sample_pair(self, level, xyz, off, ends, fn_keys, fn_down):
        """The window-key subset and the TransitionDown sample of one level from ONE farthest-point run (``_fps_pair``)."""
        keys, k_ends, down, d_ends = _fps_pair(xyz, off, ends, fn_keys, fn_down)
        to_dev = lambda v: torch.tensor(v, dtype=torch.int32).to(xyz.device, non_blocking=True)
        self.samples[("keys", level)] = (keys, to_dev(k_ends))
        self.samples[("down", level)] = (down, to_dev(d_ends))
        return keys, self.samples[("down", level)] + (d_ends,)

    def precompute(self, layers=None):
        """``layers``: the model's BasicLayers by level (``StratifiedTransformer.layers_by_level()``) -> also their window edge tables.
        Those have data-dependent shapes (host reads of counts), so that part belongs on a worker thread (StratifiedPrefetcher)."""
        scale, ratio, num_layers, stem_transformer = self.cfg
        xyz, off, ends = self.coord, self.offset, self.offset_host
        level = 0
        with torch.no_grad():
            if layers is not None:
                self._ball()
            if not stem_transformer:   # a TransitionDown follows the KPConv stem (:737-741)
                idx, n_off, ends = self._sample(("down", "stem"), xyz, off, ends, lambda n: int(n * ratio) + 1)
                n_xyz, level = xyz[idx.long(), :].contiguous(), 1
                if layers is not None:
                    self._level_neighbors("stem", xyz, off, n_xyz, n_off)
                xyz, off = n_xyz, n_off
            for l in range(level, num_layers):
                if l < num_layers - 1:   # both subsets of the level from one farthest-point run
                    keys, (idx, n_off, n_ends) = self._sample_pair(l, xyz, off, ends, lambda n: n // scale + 1, lambda n: int(n * ratio) + 1)
                else:
                    keys = self._sample(("keys", l), xyz, off, ends, lambda n: n // scale + 1)[0]
                if layers is not None:
                    self.windows[l] = layers[l].window_tables(xyz, off, keys)
                if l < num_layers - 1:
                    n_xyz = xyz[idx.long(), :].contiguous()
                    if layers is not None:
                        self._level_neighbors(l, xyz, off, n_xyz, n_off)
                    xyz, off, ends = n_xyz, n_off, n_ends
        return self

    @staticmethod
    def precompute_group(geoms, layers=None):
        """``precompute`` for several batches with ONE farthest-point chain: the scenes of all batches go through each level's sampling
        launch together (one workgroup per scene: D batches cost the latency of one), every batch then gets its own row / index slices
        (indices rebased to the batch's level arrays) and its own window tables.  Bit-identical to ``precompute`` per batch
        (tests/test_gpu_pointops2.py::test_stratified_group_prepass_is_the_per_batch_prepass)."""
        scale, ratio, num_layers, stem_transformer = geoms[0].cfg
        assert stem_transformer and all(g.cfg == geoms[0].cfg for g in geoms)
        dev = geoms[0].coord.device
        to_dev = lambda v: torch.tensor(v, dtype=torch.int32).to(dev, non_blocking=True)
        nsc = [len(g.offset_host) for g in geoms]                       # scenes per batch
        s0 = [sum(nsc[:b]) for b in range(len(geoms))]
        with torch.no_grad():
            xyz = torch.cat([g.coord for g in geoms]) if len(geoms) > 1 else geoms[0].coord
            ends, base = [], 0
            for g in geoms:
                ends += [base + e for e in g.offset_host]
                base = ends[-1]
            off = to_dev(ends)
            for l in range(num_layers):
                last = l == num_layers - 1
                if not last:
                    keys, k_ends, down, d_ends = _fps_pair(xyz, off, ends, lambda n: n // scale + 1, lambda n: int(n * ratio) + 1)
                else:
                    k_ends = _strided_counts(ends, lambda n: n // scale + 1)
                    keys = pointops.furthestsampling(xyz, off, to_dev(k_ends))
                for b, g in enumerate(geoms):
                    a, z = s0[b], s0[b] + nsc[b]                         # the batch's scenes in the group
                    p0 = ends[a - 1] if a else 0                          # first row of the batch in this level's arrays
                    local_ends = [e - p0 for e in ends[a:z]]
                    kb = k_ends[a - 1] if a else 0
                    g.samples[("keys", l)] = ((keys[kb:k_ends[z - 1]] - p0).contiguous(), to_dev([e - kb for e in k_ends[a:z]]))
                    if not last:
                        db = d_ends[a - 1] if a else 0
                        g.samples[("down", l)] = ((down[db:d_ends[z - 1]] - p0).contiguous(), to_dev([e - db for e in d_ends[a:z]]))
                    if layers is not None:
                        xb, ob = (g.coord if l == 0 else xyz[p0:ends[z - 1]]), (g.offset if l == 0 else to_dev(local_ends))
                        g.windows[l] = layers[l].window_tables(xb, ob, g.samples[("keys", l)][0])
                        if l == 0:
                            g._ball()
                        if not last:
                            d_idx, d_off = g.samples[("down", l)]
                            g._level_neighbors(l, xb, ob, xb[d_idx.long(), :].contiguous(), d_off)
                if not last:
                    xyz, ends = xyz[down.long(), :].contiguous(), d_ends
                    off = to_dev(ends)
        return geoms

    def tensors(self):
        out = [self.coord, self.offset] + [t for pair in self.samples.values() for t in pair]
        for tables in self.windows.values():
            for tab in tables.values():
                out += [t for t in tab if torch.is_tensor(t)]
                order = _native.window_order_of(tab[2])   # the queries window by window, left on the CSR offsets by the edge builder
                if order is not None:
                    out.append(order)
                csc = getattr(tab[1], _native._CSC, None)   # the key-grouped edge list cached on index_1 (_native.window_csc)
                if csc is not None:
                    out += list(csc["base"]) + [csc["perm"]] + list(csc["rel"].values()) + ([csc["order"]] if "order" in csc else [])
        for v in self.neighbors.values():
            for t in (v if isinstance(v, tuple) else (v,)):
                if torch.is_tensor(t):
                    out.append(t)
                    inv = getattr(t, _native._INV, None)   # (data_ptr, version, n, (offsets, entries, base))
                    if inv is not None:
                        out += [x for x in inv[3] if torch.is_tensor(x)]
        return out


class StratifiedPrefetcher:
    """Builds the StratifiedGeometry of upcoming batches on a worker thread with its own HIP stream while the current batch trains
    (the DataLoader-worker pattern, on the device): the FPS chain and the window partitions read coordinates only.  The worker's
    host-side waits (partition sizes are data dependent) block that thread alone.  ``get()`` joins the worker, makes the consumer
    stream wait for the side stream and registers the tables with it."""

    def __init__(self, model, windows=True):
        from concurrent.futures import ThreadPoolExecutor

        self.model, self.windows = model, windows
        self.stream = torch.cuda.Stream()
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.device = torch.cuda.current_device()

    def submit(self, batch):
        # the geometry's constructor may enqueue conversions (coord.contiguous(), offset.int() for the int64 offsets of collate_fn) on
        # the caller's stream: the event the side stream waits for is recorded AFTER them
        geom = self.model.make_geometry(batch["coord"], batch["offset"], batch.get("offset_host"))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())   # the batch's tensors were produced on the caller's stream

        def work():
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                geom.precompute(self.model.layers_by_level() if self.windows else None)
                done = torch.cuda.Event()
                done.record(self.stream)
            return done

        return geom, self.pool.submit(work)

    def submit_group(self, batches, ready=None):
        """One ticket per batch; the farthest-point chain of all of them runs as one launch sequence (StratifiedGeometry.precompute_group).
        ``ready``: the event engine.GroupedGeometryLoader recorded when the batches' tensors were handed over (on its copy stream when
        it moved them): the pre-pass waits for it AND for an event recorded here, after the conversions the geometry constructors may
        enqueue on the caller's stream."""
        geoms = [self.model.make_geometry(b["coord"], b["offset"], b.get("offset_host")) for b in batches]
        handed_over = ready
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())

        def work():
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.stream):
                if handed_over is not None:
                    self.stream.wait_event(handed_over)
                self.stream.wait_event(ready)
                StratifiedGeometry.precompute_group(geoms, self.model.layers_by_level() if self.windows else None)
                done = torch.cuda.Event()
                done.record(self.stream)
            return done

        future = self.pool.submit(work)
        return [(g, future) for g in geoms]

    @staticmethod
    def get(ticket):
        geom, future = ticket
        cur = torch.cuda.current_stream()
        cur.wait_event(future.result())
        for t in geom.tensors():
            t.record_stream(cur)
        return geom

    def close(self):
        self.pool.shutdown(wait=True)


_ACTIVE_GEOMETRY = None   # the StratifiedGeometry of the forward in progress (set by StratifiedTransformer.forward)


def _fps(key, xyz, offset, fn):
    """(sample indices, scene ends of the sample): from the forward's StratifiedGeometry when it holds them, else computed here."""
    g = _ACTIVE_GEOMETRY
    if g is not None and key in g.samples:
        return g.samples[key]
    new_offset = _strided_offsets(offset, fn)
    return pointops.furthestsampling(xyz, offset.int(), new_offset), new_offset


class TransitionDown(nn.Module):
    """:156-189 -- FPS (ratio * n + 1 points per scene), kNN grouping WITHOUT coordinates, LayerNorm, Linear, max over k."""

    level = None   # set by StratifiedTransformer: key of this module's FPS call in a StratifiedGeometry

    def __init__(self, in_channels, out_channels, ratio, k, norm_layer=dense.LayerNorm):
        super().__init__()
        self.ratio, self.k = ratio, k
        self.norm = norm_layer(in_channels) if norm_layer else None
        self.linear = _Linear(in_channels, out_channels, bias=False)
        self.pool = nn.MaxPool1d(k)

    def forward(self, feats, xyz, offset):
        idx, n_offset = _fps(("down", self.level), xyz, offset, lambda n: int(n * self.ratio) + 1)
        g = _ACTIVE_GEOMETRY
        n_xyz, knn_idx = g.neighbors.get(("td", self.level), (None, None)) if g is not None else (None, None)
        if n_xyz is None:
            n_xyz = xyz[idx.long(), :].contiguous()
        feats = pointops.queryandgroup(self.k, xyz, n_xyz, feats.contiguous(), knn_idx, offset, n_offset, use_xyz=False)   # (m, k, c)
        m, k, c = feats.shape
        feats = self.linear(self.norm(feats.view(m * k, c)).view(m, k, c))
        # MaxPool1d(k) over the k neighbours (:186-188) as a reduction over dim 1: same values, no (m, c, k) transpose copy, and the
        # backward is an index scatter instead of max_pool_backward_nchw (1 ms per call at level 0: 3 ms of the step)
        return feats.max(dim=1)[0], n_xyz, n_offset


class SwinTransformerBlock(nn.Module):
    """:353-410"""

    def __init__(self, dim, num_heads, window_size, quant_size, rel_query=True, rel_key=False, rel_value=False, drop_path=0.0,
                 mlp_ratio=4.0, qkv_bias=True, qk_scale=None, act_layer=nn.GELU, norm_layer=dense.LayerNorm, mode=4):
        super().__init__()
        self.mode = mode
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size, num_heads=num_heads, quant_size=quant_size, rel_query=rel_query, rel_key=rel_key,
                                    rel_value=rel_value, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max, rel_idx=None):
        short_cut = feats
        feats = self.attn(self.norm1(feats), xyz, index_0, index_1, index_0_offsets, n_max, rel_idx)   # index_0 in ascending order
        feats = short_cut + self.drop_path(feats)
        return feats + self.drop_path(self.mlp(self.norm2(feats)))


class BasicLayer(nn.Module):
    """:413-555 -- window partition (fine + shifted, coarse + shifted), FPS key subset, ``depth`` Swin blocks, optional TransitionDown."""

    def __init__(self, downsample_scale, depth, channel, num_heads, window_size, grid_size, quant_size, rel_query=True, rel_key=False,
                 rel_value=False, drop_path=0.0, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, norm_layer=dense.LayerNorm, downsample=None,
                 ratio=0.25, k=16, out_channels=None):
        super().__init__()
        self.level = None   # set by StratifiedTransformer (key of this layer's FPS calls in a StratifiedGeometry)
        self.depth, self.grid_size, self.max_window_counts = depth, grid_size, 64
        self.window_size, self.downsample_scale = window_size, downsample_scale
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(channel, num_heads, window_size, quant_size, rel_query=rel_query, rel_key=rel_key, rel_value=rel_value,
                                 drop_path=(drop_path[i] if isinstance(drop_path, list) else drop_path), mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(channel, out_channels, ratio, k) if downsample else None

    def window_tables(self, xyz, offset, downsample_idx):
        """The edge tables of this layer's two window partitions (even blocks: plain, odd blocks: shifted by half a window) -- :468-536,
        which rebuilds them for every block although they depend on the coordinates and the block's parity only.  Per parity:
        (index_0 sorted, index_1, CSR offsets of index_0, longest row, relative-position table rows of the first block's attention)."""
        xyz_min, xyz_max = _colminmax(xyz)
        be = _native.backend_for(xyz)
        on_device = hasattr(be, "window_keys") and xyz.dtype == torch.float32
        if not on_device:
            window_size = torch.tensor([self.window_size] * 3, dtype=xyz.dtype, device=xyz.device)
            batch = offset2batch(offset, xyz.shape[0])
        tables, flags = {}, []
        for parity in range(2 if self.depth > 1 else 1):
            if on_device:   # one launch (csrc/window_edges.hip we::k_keys): the same float32 steps as the torch composition below
                kf, kc, wk = be.window_keys(xyz, offset, xyz_min, xyz_max, self.window_size, parity)
            else:
                kf, kc, wk = window_keys(xyz, batch, window_size, xyz_min, parity)
            attn = self.blocks[parity].attn
            # every query's row = [its fine window, ascending] ++ [the downsampled points of its coarse window in another fine window,
            # ascending] -- what :45-100 + the stable sort by query of :507 produce -- with the quantised relative positions of :282-292
            index_0, index_1, index_0_offsets, n_max, rel_idx, flag = be.window_edges(
                xyz, kf, kc, wk, downsample_idx, 2 * attn.window_size, attn.quant_size, 2 * attn.quant_grid_length - 1)
            flags.append(flag)
            if xyz.is_cuda:   # the backward's key-grouped edge list: coordinate-only like everything else here, cached on index_1
                _native.window_csc(index_1, index_0_offsets, rel_idx, n_keys=xyz.shape[0])
            tables[parity] = (index_0, index_1, index_0_offsets, n_max, rel_idx)
        # WindowAttention asserts 0 <= rel_idx < 2 * quant_grid_length upstream (:291): one host read for the layer's partitions
        assert not bool(torch.stack(flags).any()), "window edge tables: a quantised relative position left the table (coordinates outside the window?)"
        return tables

    def forward(self, feats, xyz, offset):
        g = _ACTIVE_GEOMETRY
        tables = g.windows.get(self.level) if g is not None else None
        if tables is None:
            downsample_idx, _ = _fps(("keys", self.level), xyz, offset, lambda n: n // self.downsample_scale + 1)
            tables = self.window_tables(xyz, offset, downsample_idx)
        for i, blk in enumerate(self.blocks):
            feats = blk(feats, xyz, *tables[i % 2])
        if self.downsample:
            feats_down, xyz_down, offset_down = self.downsample(feats, xyz, offset)
        else:
            feats_down, xyz_down, offset_down = None, None, None
        return feats, xyz, offset, feats_down, xyz_down, offset_down


class Upsample(nn.Module):
    """:558-579 (also recognizer_model/st_v1m1.py:6-27)"""

    def __init__(self, k, in_channels, out_channels, bn_momentum=0.02):
        super().__init__()
        self.k, self.in_channels, self.out_channels = k, in_channels, out_channels
        self.linear1 = nn.Sequential(dense.LayerNorm(out_channels), _Linear(out_channels, out_channels))
        self.linear2 = nn.Sequential(dense.LayerNorm(in_channels), _Linear(in_channels, out_channels))

    level = None   # set by the owner: the level this module interpolates ONTO (key of its tables in a StratifiedGeometry)

    def forward(self, feats, xyz, support_xyz, offset, support_offset, support_feats=None):
        g = _ACTIVE_GEOMETRY
        tab = g.neighbors.get(("up", self.level)) if g is not None and self.level is not None else None
        coarse = self.linear2(feats).contiguous()
        if tab is not None and tab[0].shape[0] == support_xyz.shape[0] and coarse.dtype == torch.float32:
            up = _p1_ops._InterpolateIdx.apply(coarse, tab[0], tab[1])
        else:
            up = pointops.interpolation(xyz.contiguous(), support_xyz.contiguous(), coarse, offset, support_offset)
        return self.linear1(support_feats) + up, support_xyz, support_offset


class KPConvSimpleBlock(nn.Module):
    """:582-607"""

    def __init__(self, in_channels, out_channels, prev_grid_size, sigma=1.0, negative_slope=0.2, bn_momentum=0.02):
        super().__init__()
        self.kpconv = KPConvLayer(in_channels, out_channels, point_influence=prev_grid_size * sigma, add_one=False)
        self.bn = FastBatchNorm1d(out_channels, momentum=bn_momentum)
        self.activation = nn.LeakyReLU(negative_slope=negative_slope)

    def forward(self, feats, xyz, batch, neighbor_idx):
        return self.activation(self.bn(self.kpconv(xyz, xyz, neighbor_idx, feats)))


class KPConvResBlock(nn.Module):
    """:610-662"""

    def __init__(self, in_channels, out_channels, prev_grid_size, sigma=1.0, negative_slope=0.2, bn_momentum=0.02):
        super().__init__()
        d_2 = out_channels // 4
        activation = nn.LeakyReLU(negative_slope=negative_slope)
        self.unary_1 = nn.Sequential(_Linear(in_channels, d_2, bias=False), FastBatchNorm1d(d_2, momentum=bn_momentum), activation)
        self.unary_2 = nn.Sequential(_Linear(d_2, out_channels, bias=False), FastBatchNorm1d(out_channels, momentum=bn_momentum), activation)
        self.kpconv = KPConvLayer(d_2, d_2, point_influence=prev_grid_size * sigma, add_one=False)
        self.bn = FastBatchNorm1d(out_channels, momentum=bn_momentum)
        self.activation = activation
        if in_channels != out_channels:
            self.shortcut_op = nn.Sequential(_Linear(in_channels, out_channels, bias=False), FastBatchNorm1d(out_channels, momentum=bn_momentum))
        else:
            self.shortcut_op = nn.Identity()

    def forward(self, feats, xyz, batch, neighbor_idx):
        shortcut = feats
        feats = self.unary_2(self.kpconv(xyz, xyz, neighbor_idx, self.unary_1(feats)))
        return feats + self.shortcut_op(shortcut)


def _set_upsample_levels(upsamples, num_layers, stem_transformer=True):
    """Upsample i of the stack interpolates from level num_layers - 1 - i onto the level below it ("stem": the full-resolution points in
    front of the first TransitionDown of the stem_transformer=False variant)."""
    for i, up in enumerate(upsamples):
        target = num_layers - 2 - i
        up.level = target if (stem_transformer or target > 0) else "stem"


@MODELS.register_module("ST-v1m1")
class StratifiedTransformer(nn.Module):
    """:665-845"""

    def __init__(self, downsample_scale, depths, channels, num_heads, window_size, up_k, grid_sizes, quant_sizes, rel_query=True,
                 rel_key=False, rel_value=False, drop_path_rate=0.2, num_layers=4, concat_xyz=False, num_classes=13, ratio=0.25, k=16,
                 prev_grid_size=0.04, sigma=1.0, stem_transformer=False, kp_ball_radius=0.02 * 2.5, kp_max_neighbor=34):
        super().__init__()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]   # stochastic depth decay rule
        self.kp_ball_radius, self.kp_max_neighbor = kp_ball_radius, kp_max_neighbor
        cin = 3 if not concat_xyz else 6
        if stem_transformer:
            self.stem_layer = nn.ModuleList([KPConvSimpleBlock(cin, channels[0], prev_grid_size, sigma=sigma)])
            self.layer_start = 0
        else:
            self.stem_layer = nn.ModuleList([KPConvSimpleBlock(cin, channels[0], prev_grid_size, sigma=sigma),
                                             KPConvResBlock(channels[0], channels[0], prev_grid_size, sigma=sigma)])
            self.downsample = TransitionDown(channels[0], channels[1], ratio, k)
            self.layer_start = 1
        self.layers = nn.ModuleList([
            BasicLayer(downsample_scale, depths[i], channels[i], num_heads[i], window_size[i], grid_sizes[i], quant_sizes[i],
                       rel_query=rel_query, rel_key=rel_key, rel_value=rel_value, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                       downsample=TransitionDown if i < num_layers - 1 else None, ratio=ratio, k=k,
                       out_channels=channels[i + 1] if i < num_layers - 1 else None)
            for i in range(self.layer_start, num_layers)])
        for j, layer in enumerate(self.layers):
            layer.level = self.layer_start + j
            if layer.downsample is not None:
                layer.downsample.level = self.layer_start + j
        if not stem_transformer:
            self.downsample.level = "stem"
        self.geometry_cfg = dict(downsample_scale=downsample_scale, ratio=ratio, num_layers=num_layers, stem_transformer=stem_transformer, k=k,
                                 ball=(kp_ball_radius, kp_max_neighbor))
        self.upsamples = nn.ModuleList([Upsample(up_k, channels[i], channels[i - 1]) for i in range(num_layers - 1, 0, -1)])
        _set_upsample_levels(self.upsamples, num_layers, stem_transformer)
        self.classifier = nn.Sequential(_Linear(channels[0], channels[0]), nn.BatchNorm1d(channels[0]), nn.ReLU(inplace=True),
                                        _Linear(channels[0], num_classes))
        self.init_weights()

    def layers_by_level(self):
        return {layer.level: layer for layer in self.layers}

    def make_geometry(self, coord, offset, offset_host=None):
        """The batch's StratifiedGeometry (not yet computed): ``.precompute()`` it on any stream ahead of the step and pass it as
        ``data_dict["st_geometry"]``."""
        return StratifiedGeometry(coord, offset, offset_host, **self.geometry_cfg)

    def forward(self, data_dict):
        global _ACTIVE_GEOMETRY
        _ACTIVE_GEOMETRY = data_dict.get("st_geometry") if isinstance(data_dict, dict) else None
        try:
            return self._forward(data_dict)
        finally:
            _ACTIVE_GEOMETRY = None

    def _forward(self, data_dict):
        from .pseudo_label import radius_neighbors

        feats, xyz, offset = data_dict["feat"], data_dict["coord"].contiguous(), data_dict["offset"].int()
        batch = offset2batch(offset, xyz.shape[0])
        # tp.ball_query(radius, max_neighbor, xyz, xyz, mode="partial_dense", batch_x, batch_y)[0]  (:766-774)
        g = _ACTIVE_GEOMETRY
        neighbor_idx = g.neighbors.get(("ball",)) if g is not None and g.nn_cfg[2] == (self.kp_ball_radius, self.kp_max_neighbor) else None
        if neighbor_idx is None:
            neighbor_idx = radius_neighbors(xyz, offset, self.kp_ball_radius, self.kp_max_neighbor)
        feats_stack, xyz_stack, offset_stack = [], [], []
        for layer in self.stem_layer:
            feats = layer(feats, xyz, batch, neighbor_idx)
        feats = feats.contiguous()
        if self.layer_start == 1:
            feats_stack.append(feats); xyz_stack.append(xyz); offset_stack.append(offset)
            feats, xyz, offset = self.downsample(feats, xyz, offset)
        for layer in self.layers:
            feats, xyz, offset, feats_down, xyz_down, offset_down = layer(feats, xyz, offset)
            feats_stack.append(feats); xyz_stack.append(xyz); offset_stack.append(offset)
            feats, xyz, offset = feats_down, xyz_down, offset_down
        feats, xyz, offset = feats_stack.pop(), xyz_stack.pop(), offset_stack.pop()
        for upsample in self.upsamples:
            feats, xyz, offset = upsample(feats, xyz, xyz_stack.pop(), offset, offset_stack.pop(), support_feats=feats_stack.pop())
        return self.classifier(feats)

    def init_weights(self):
        """:832-845"""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

        self.apply(_init)


@MODELS.register_module("ST-v1m1-Recognizer")
class STRecognizer(nn.Module):
    """PDF U-decoder over the hooked inputs / outputs of the backbone's four-level Upsample stack -- recognizer_model/st_v1m1.py:30-69."""

    def __init__(self, up_k, channels, num_layers):
        super().__init__()
        self.upsamples = nn.ModuleList([Upsample(up_k, channels[i], channels[i - 1]) for i in range(num_layers - 1, 0, -1)])
        _set_upsample_levels(self.upsamples, num_layers)
        self.confidence = nn.Sequential(_Linear(channels[0], channels[0]), nn.BatchNorm1d(channels[0]), nn.ReLU(inplace=True),
                                        _Linear(channels[0], 1))

    def forward(self, model_hooks):
        n = len(self.upsamples)
        in_feats = [model_hooks[f"backbone.upsamples.{i}"]["forward_input"] for i in range(n)]
        out_feats = [model_hooks[f"backbone.upsamples.{i}"]["forward_output"] for i in range(n)]
        feats = in_feats[0][0]
        for i, upsample in enumerate(self.upsamples):
            feats, _, _ = upsample(feats, in_feats[i][1], in_feats[i][2], in_feats[i][3], in_feats[i][4], out_feats[i][0])
        return self.confidence(feats)
