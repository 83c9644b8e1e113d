"""CPU suite for SURVEY 8 row f-1 (libs/pointops2 window attention): the host wrappers over the oracle reproduce the fixtures
produced by the REFERENCE's own autograd wrappers (tests/golden/ops_pointops2_ref.npz, made by tests/golden/make_golden.py from
libs/pointops2/functions/pointops.py), and the oracle's restatement of the v2 / v3 CUDA kernels agrees with the dense
edge-list formulas -- the design of the reference's own v1-vs-v2 scripts (libs/pointops2/functions/test_attention_op_step1_v2.py,
test_relative_pos_encoding_op_step1_v3.py, test_relative_pos_encoding_op_step2_v2.py), with torch autograd as the second side."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from pointcloudpdf_amd.pointops2 import pointops as p2  # noqa: E402


def window_graph(seed, n, h, d, L, max_deg):
    """Same generator as tests/golden/make_golden.py::window_graph (inputs are regenerated, only outputs are stored)."""
    g = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, max_deg + 1, (n,), generator=g)
    deg[::7] = 0
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)]).int()
    m = int(offsets[-1])
    index1 = torch.randint(0, n, (m,), generator=g).int()
    rel_idx = torch.randint(0, L, (m, 3), generator=g).int()
    q, k, v = (torch.randn(n, h, d, generator=g) for _ in range(3))
    tq, tk, tv = (torch.randn(L, h, d, 3, generator=g) * 0.5 for _ in range(3))
    return dict(offsets=offsets, index1=index1, rel_idx=rel_idx, q=q, k=k, v=v, tq=tq, tk=tk, tv=tv, n_max=int(deg.max()), m=m)


WINDOW_CASES = {"h3d16": (11, 300, 3, 16, 24, 40), "h2d32": (12, 170, 2, 32, 10, 90)}


def chain(ops, G, dev="cpu"):
    """attn -> bias -> value aggregation as in WindowAttention.forward (stratified_transformer_v1m1_origin.py:277-341)."""
    T = lambda t: t.to(dev)
    q, k, v = (T(G[n]).clone().requires_grad_(True) for n in ("q", "k", "v"))
    tq, tk, tv = (T(G[n]).clone().requires_grad_(True) for n in ("tq", "tk", "tv"))
    off, i1, rel = T(G["offsets"]), T(G["index1"]), T(G["rel_idx"])
    attn = ops.attention_step1_v2(q, k, i1, off, G["n_max"])
    bias = ops.dot_prod_with_idx_v3(q, off, G["n_max"], k, i1, tq, tk, rel)
    x = ops.attention_step2_with_rel_pos_value_v2((attn + bias) * 0.1, v, off, G["n_max"], i1, tv, rel)
    gx = torch.randn(x.shape, generator=torch.Generator().manual_seed(99)).to(dev)
    x.backward(gx)
    res = dict(attn=attn, bias=bias, x=x, gx=gx)
    res.update({"g" + nm: t.grad for nm, t in (("q", q), ("k", k), ("v", v), ("tq", tq), ("tk", tk), ("tv", tv))})
    return {k_: v_.detach().cpu() for k_, v_ in res.items()}


class DenseEdgeList:
    """The same three ops written with plain torch indexing over the expanded edge list (index0 = query of every edge): the
    'v1' semantics the reference's test scripts compare v2 / v3 against."""

    @staticmethod
    def _index0(off):
        deg = (off[1:] - off[:-1]).long()
        return torch.repeat_interleave(torch.arange(deg.shape[0], device=off.device), deg)

    @staticmethod
    def _table(t, rel):
        r = rel.long()
        return t[r[:, 0], :, :, 0] + t[r[:, 1], :, :, 1] + t[r[:, 2], :, :, 2]   # (M, h, d)

    @classmethod
    def attention_step1_v2(cls, q, k, index1, off, n_max):
        return (q[cls._index0(off)] * k[index1.long()]).sum(-1)

    @classmethod
    def dot_prod_with_idx_v3(cls, q, off, n_max, k, index_k, tq, tk, rel):
        return (q[cls._index0(off)] * cls._table(tq, rel)).sum(-1) + (k[index_k.long()] * cls._table(tk, rel)).sum(-1)

    @classmethod
    def attention_step2_with_rel_pos_value_v2(cls, attn, v, off, n_max, index1, table, rel):
        contrib = (v[index1.long()] + cls._table(table, rel)) * attn.unsqueeze(-1)
        out = torch.zeros_like(v)
        return out.index_add(0, cls._index0(off), contrib)


@pytest.fixture(scope="module")
def g2(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_pointops2_ref.npz"))


@pytest.mark.parametrize("tag", sorted(WINDOW_CASES))
def test_wrappers_match_reference_wrappers(use_oracle, g2, tag):
    res = chain(p2, window_graph(*WINDOW_CASES[tag]))
    for key, val in res.items():
        assert_close(val, g2[f"{tag}_{key}"], 1e-6, f"{tag} {key}")


@pytest.mark.parametrize("tag", sorted(WINDOW_CASES))
def test_oracle_restatement_matches_dense_edge_list_formulas(use_oracle, tag):
    G = window_graph(*WINDOW_CASES[tag])
    ours, dense = chain(p2, G), chain(DenseEdgeList, G)
    for key in ours:
        assert_close(ours[key], dense[key], 2e-5, f"{tag} {key}")


def test_pointops2_names_and_adapters(use_oracle):
    for n in ("attention_step1_v2", "dot_prod_with_idx_v3", "attention_step2_with_rel_pos_value_v2", "furthestsampling", "knnquery",
              "interpolation"):
        assert callable(getattr(p2, n)), n
    import pointcloudpdf_amd.pointops2 as pkg
    assert pkg.pointops is p2
    xyz = torch.rand(200, 3)
    off = torch.tensor([120, 200], dtype=torch.int32)
    idx, dist = p2.knnquery(4, xyz, None, off, off)   # pointops2 order: (nsample, xyz, new_xyz, offset, new_offset)
    from pointcloudpdf_amd import pointops as p1
    ref_idx, ref_dist = p1.knn_query(4, xyz, off)
    assert torch.equal(idx, ref_idx) and torch.equal(dist, ref_dist)
    noff = torch.tensor([30, 50], dtype=torch.int32)
    assert torch.equal(p2.furthestsampling(xyz, off, noff), p1.farthest_point_sampling(xyz, off, noff))


def test_queryandgroup_matches_reference_composition(use_oracle):
    """libs/pointops2/functions/pointops.py:964-1001 written out with plain indexing (scenes with at least nsample points)."""
    g = torch.Generator().manual_seed(3)
    xyz = torch.rand(260, 3, generator=g)
    feat = torch.randn(260, 5, generator=g)
    off = torch.tensor([100, 260], dtype=torch.int32)
    new_xyz = xyz[::4].contiguous()
    noff = torch.tensor([25, 65], dtype=torch.int32)
    for use_xyz in (True, False):
        out, idx = p2.queryandgroup(8, xyz, new_xyz, feat, None, off, noff, use_xyz=use_xyz, return_indx=True)
        flat = idx.view(-1).long()
        gx = xyz[flat].view(65, 8, 3) - new_xyz.unsqueeze(1)
        gf = feat[flat].view(65, 8, 5)
        want = torch.cat((gx, gf), -1) if use_xyz else gf
        assert torch.equal(out, want)
    assert p2.queryandgroup(8, xyz, new_xyz, feat, idx, off, noff, use_xyz=False).shape == (65, 8, 5)


def edge_list_case(seed=5, n=150, h=3, d=16, L=12, m=2000):
    g = torch.Generator().manual_seed(seed)
    G = dict(index0=torch.randint(0, n, (m,), generator=g).int(), index1=torch.randint(0, n, (m,), generator=g).int(),
             rel_idx=torch.randint(0, L, (m, 3), generator=g).int(), attn=torch.randn(m, h, generator=g))
    G["index0"][0] = n - 1   # index0.max() + 1 == n (upstream sizes outputs that way)
    for nm in ("q", "k", "v"):
        G[nm] = torch.randn(n, h, d, generator=g)
    for nm in ("tq", "tk", "tv"):
        G[nm] = torch.randn(L, h, d, 3, generator=g) * 0.5
    return G


def run_edge_list_ops(ops, G, dev="cpu"):
    """The v1 edge-list forms on an UNSORTED edge list, forward values and gradients of every float input."""
    T = lambda t: t.to(dev)
    leaf = {nm: T(G[nm]).clone().requires_grad_(True) for nm in ("q", "k", "v", "tq", "tk", "tv", "attn")}
    i0, i1, rel = T(G["index0"]), T(G["index1"]), T(G["rel_idx"])
    outs = dict(
        step1=ops.attention_step1(leaf["q"], leaf["k"], i0, i1),
        dot=ops.dot_prod_with_idx(leaf["q"], i0, leaf["tq"], rel),
        dot2=ops.dot_prod_with_idx_v2(leaf["q"], i0, leaf["k"], i1, leaf["tq"], leaf["tk"], rel),
        step2=ops.attention_step2(leaf["attn"], leaf["v"], i0, i1),
        step2rv=ops.attention_step2_with_rel_pos_value(leaf["attn"], leaf["v"], i0, i1, leaf["tv"], rel),
    )
    total = sum((o * torch.cos(torch.arange(o.numel(), device=o.device, dtype=torch.float32)).view_as(o)).sum() for o in outs.values())
    total.backward()
    res = {k_: v_.detach().cpu() for k_, v_ in outs.items()}
    res.update({"g_" + nm: t.grad.detach().cpu() for nm, t in leaf.items()})
    return res


class DenseEdgeListV1:
    @staticmethod
    def _table(t, rel):
        r = rel.long()
        return t[r[:, 0], :, :, 0] + t[r[:, 1], :, :, 1] + t[r[:, 2], :, :, 2]

    @staticmethod
    def attention_step1(q, k, index0, index1):
        return (q[index0.long()] * k[index1.long()]).sum(-1)

    @classmethod
    def dot_prod_with_idx(cls, q, index, table, rel):
        return (q[index.long()] * cls._table(table, rel)).sum(-1)

    @classmethod
    def dot_prod_with_idx_v2(cls, q, index_q, k, index_k, tq, tk, rel):
        return (q[index_q.long()] * cls._table(tq, rel)).sum(-1) + (k[index_k.long()] * cls._table(tk, rel)).sum(-1)

    @staticmethod
    def attention_step2(attn, v, index0, index1):
        nq = int(index0.max()) + 1
        return torch.zeros(nq, *v.shape[1:], dtype=v.dtype, device=v.device).index_add(0, index0.long(), v[index1.long()] * attn.unsqueeze(-1))

    @classmethod
    def attention_step2_with_rel_pos_value(cls, attn, v, index0, index1, table, rel):
        nq = int(index0.max()) + 1
        contrib = (v[index1.long()] + cls._table(table, rel)) * attn.unsqueeze(-1)
        return torch.zeros(nq, *v.shape[1:], dtype=v.dtype, device=v.device).index_add(0, index0.long(), contrib)


def test_v1_edge_list_forms_match_dense_formulas(use_oracle):
    """libs/pointops2/functions/pointops.py:93-167, 261-335, 407-473, 476-629, 758-851 on an unsorted edge list."""
    G = edge_list_case()
    ours, dense = run_edge_list_ops(p2, G), run_edge_list_ops(DenseEdgeListV1, G)
    assert set(ours) == set(dense)
    for key in ours:
        assert_close(ours[key], dense[key], 2e-5, key)


def dense_segment_softmax(x, off):
    idx0 = DenseEdgeList._index0(off)
    n = off.shape[0] - 1
    mx = torch.full((n, x.shape[1]), -3.0e38, dtype=x.dtype, device=x.device).scatter_reduce(0, idx0[:, None].expand_as(x), x, "amax")
    e = torch.exp(x - mx[idx0])
    return e / torch.zeros_like(mx).index_add(0, idx0, e)[idx0]


def test_segment_softmax_matches_dense_formula(use_oracle):
    G = window_graph(*WINDOW_CASES["h3d16"])
    x = torch.randn(G["m"], 3, generator=torch.Generator().manual_seed(8)).requires_grad_(True)
    y = p2.segment_softmax(x, G["offsets"])
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    yr = dense_segment_softmax(xr, G["offsets"])
    yr.backward(gy)
    assert_close(y, yr, 1e-6, "segment softmax")
    assert_close(x.grad, xr.grad, 1e-5, "segment softmax grad")


class DenseWindowAttention(torch.nn.Module):
    """WindowAttention.forward (stratified_transformer_v1m1_origin.py:253-350) written with dense edge-list indexing, sharing the
    parameters of the module under test."""

    def __init__(self, mod):
        super().__init__()
        self.m = mod

    def forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max):
        m = self.m
        n, c = feats.shape
        qkv = m.qkv(feats).reshape(n, 3, m.num_heads, c // m.num_heads).permute(1, 0, 2, 3)
        q, k, v = qkv[0] * m.scale, qkv[1], qkv[2]
        i0, i1 = index_0.long(), index_1.long()
        rel = m.relative_position_index(xyz, index_0, index_1).long()
        tab = lambda t: t[rel[:, 0], :, :, 0] + t[rel[:, 1], :, :, 1] + t[rel[:, 2], :, :, 2]
        attn = (q[i0] * k[i1]).sum(-1) + (q[i0] * tab(m.relative_pos_query_table)).sum(-1) + (k[i1] * tab(m.relative_pos_key_table)).sum(-1)
        attn = dense_segment_softmax(attn, index_0_offsets)
        x = torch.zeros_like(v).index_add(0, i0, (v[i1] + tab(m.relative_pos_value_table)) * attn.unsqueeze(-1))
        return m.proj(x.reshape(n, c))


def window_attention_case(dev="cpu", seed=3, n=400, dim=48, heads=3, max_deg=30):
    from pointcloudpdf_amd.stratified import WindowAttention

    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * 0.3          # inside one 0.4 m window: |offset| < window_size
    deg = torch.randint(1, max_deg + 1, (n,), generator=g)
    off = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)]).int()
    m = int(off[-1])
    index_0 = torch.repeat_interleave(torch.arange(n), deg)
    index_1 = torch.randint(0, n, (m,), generator=g)
    feats = torch.randn(n, dim, generator=g)
    torch.manual_seed(seed)
    mod = WindowAttention(dim, window_size=0.4, num_heads=heads, quant_size=0.05, rel_query=True, rel_key=True, rel_value=True)
    with torch.no_grad():
        for t in (mod.relative_pos_query_table, mod.relative_pos_key_table, mod.relative_pos_value_table):
            t.mul_(10.0)                              # trunc_normal(std=0.02) tables would hide table errors behind the qk term
    T = lambda t: t.to(dev)
    return mod.to(dev), [T(feats), T(xyz), T(index_0), T(index_1), T(off), int(deg.max())]


def run_window_attention(mod, dense, args):
    outs = []
    for net in (mod, dense):
        for p in mod.parameters():
            p.grad = None
        f = args[0].clone().requires_grad_(True)
        y = net(f, *args[1:])
        y.backward(torch.cos(torch.arange(y.numel(), device=y.device, dtype=torch.float32)).view_as(y))
        outs.append(dict(y=y.detach().cpu(), gf=f.grad.cpu(), **{n_: p.grad.detach().cpu().clone() for n_, p in mod.named_parameters()}))
    return outs


def test_window_attention_module_matches_dense_restatement(use_oracle):
    mod, args = window_attention_case()
    assert set(n_ for n_, _ in mod.named_parameters()) == {"relative_pos_query_table", "relative_pos_key_table", "relative_pos_value_table",
                                                           "qkv.weight", "qkv.bias", "proj.weight", "proj.bias"}   # reference checkpoint keys
    ours, dense = run_window_attention(mod, DenseWindowAttention(mod), args)
    for key in ours:
        assert_close(ours[key], dense[key], 5e-5, key)
