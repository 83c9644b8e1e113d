"""``WindowAttention`` of StratifiedTransformer (pointcept/models/stratified_transformer/stratified_transformer_v1m1_origin.py:185-350)
on the window-attention kernels of ``pointcloudpdf_amd.pointops2`` -- the caller side of SURVEY.md 8 row f-1.

Same constructor arguments, parameter names (``qkv``, ``proj``, ``relative_pos_{query,key,value}_table``) and forward signature as the
reference class, so its checkpoints load; what differs: ``scatter_softmax`` (torch_scatter, absent here) is the CSR segment softmax
kernel, ``trunc_normal_`` comes from ``torch.nn.init`` (timm is absent), and the non-default table modes use the edge-list forms.
The rest of the model (KPConv stem of torch_points3d, window / voxel grouping of torch_points_kernels and torch_geometric) depends
on unvendored packages and is not rebuilt.
"""
import torch
import torch.nn as nn

from .pointops2 import pointops


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
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop, inplace=True)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop, inplace=True)

    def relative_position_index(self, xyz, index_0, index_1):
        """:282-292 -- quantised offset of every edge, (M, 3) in [0, 2 * quant_grid_length)"""
        rel = xyz[index_0.long()] - xyz[index_1.long()]
        rel = torch.round(rel * 100000) / 100000
        return torch.div(rel + 2 * self.window_size - 1e-4, self.quant_size, rounding_mode="trunc")

    def forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max):
        n, c = feats.shape
        assert index_0.shape[0] == index_1.shape[0]
        qkv = self.qkv(feats).reshape(n, 3, self.num_heads, c // self.num_heads).permute(1, 0, 2, 3).contiguous()
        query, key, value = qkv[0], qkv[1], qkv[2]
        query = query * self.scale
        i1, off = index_1.int().contiguous(), index_0_offsets.int().contiguous()
        attn = pointops.attention_step1_v2(query.float(), key.float(), i1, off, n_max)
        rel_idx = self.relative_position_index(xyz, index_0, index_1)
        assert (rel_idx >= 0).all() and (rel_idx <= 2 * self.quant_grid_length - 1).all()
        rel_idx = rel_idx.int().contiguous()
        if self.rel_query and self.rel_key:
            bias = pointops.dot_prod_with_idx_v3(query.float(), off, n_max, key.float(), i1, self.relative_pos_query_table.float(),
                                                 self.relative_pos_key_table.float(), rel_idx)
        elif self.rel_query:
            bias = pointops.dot_prod_with_idx(query.float(), index_0.int(), self.relative_pos_query_table.float(), rel_idx)
        elif self.rel_key:
            bias = pointops.dot_prod_with_idx(key.float(), index_1.int(), self.relative_pos_key_table.float(), rel_idx)
        else:
            bias = 0.0
        attn = pointops.segment_softmax(attn + bias, off)                                  # scatter_softmax(src, index_0, dim=0), :322-324
        # (upstream constructs ``attn_drop`` but never applies it -- :246 vs :322-341 -- so neither does this forward)
        if self.rel_value:
            x = pointops.attention_step2_with_rel_pos_value_v2(attn.float(), value.float(), off, n_max, i1,
                                                               self.relative_pos_value_table.float(), rel_idx)
        else:
            x = pointops.attention_step2(attn.float(), value.float(), index_0.int(), index_1.int())
        x = self.proj(x.view(n, c))
        return self.proj_drop(x)
