"""CPU suite: StratifiedTransformer (ST-v1m1) + ST-v1m1-Recognizer host code on the CPU oracle against the fixture produced by the
reference's OWN classes (tests/golden/model_stratified.npz; stratified_transformer_v1m1_origin.py, recognizer_model/st_v1m1.py)."""
import os

import numpy as np
import pytest
import torch

import helpers
from pointcloudpdf_amd import stratified
from pointcloudpdf_amd.registry import MODELS


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_stratified_model_matches_reference_classes(use_oracle, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "model_stratified.npz"))
    torch.manual_seed(0)
    out = helpers.run_stratified_case(mode)
    helpers.check_stratified_case(mode, out, g)


@pytest.mark.parametrize("geometry", ["fps", "windows"])
def test_stratified_geometry_ahead_of_the_forward_changes_nothing(use_oracle, geometry):
    """A StratifiedGeometry computed before the forward (FPS chain only, or with the window edge tables) feeds the same indices the
    forward would compute itself: logits, confidence and loss are bit-identical; gradients to the run-to-run noise of the CPU
    backward's threaded accumulation (1e-6 between two plain runs)."""
    a = helpers.run_stratified_case("train")
    b = helpers.run_stratified_case("train", geometry=geometry)
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["conf"], b["conf"]) and torch.equal(a["loss"], b["loss"])
    ga, gb = dict(a["model"].named_parameters()), dict(b["model"].named_parameters())
    for k in ga:
        if ga[k].grad is not None:
            assert (ga[k].grad - gb[k].grad).abs().max() <= 1e-5 * (ga[k].grad.abs().max() + 1e-30), k
    assert len(b["model"].backbone.layers_by_level()) == 4
    # the full pre-pass also holds the forward's neighbour searches (KPConv radius table, TransitionDown kNN, Upsample interpolation tables)
    want = {("ball",)} | {(kind, l) for kind in ("td", "up") for l in range(3)} if geometry == "windows" else set()
    assert set(b["geometry"].neighbors) == want


def test_stratified_state_dict_layout_and_registry():
    """Parameter names of the reference classes (checkpoint compatibility) and the registry names of the reference config."""
    assert "ST-v1m1" in MODELS and "ST-v1m1-Recognizer" in MODELS
    m = MODELS.build(dict(type="ST-v1m1", drop_path_rate=0.3, **helpers.ST_CFG))
    keys = set(m.state_dict().keys())
    for k in ["stem_layer.0.kpconv.weight", "stem_layer.0.bn.batch_norm.running_mean", "layers.0.blocks.0.norm1.weight",
              "layers.0.blocks.0.attn.qkv.bias", "layers.0.blocks.0.attn.relative_pos_query_table", "layers.0.blocks.1.attn.relative_pos_key_table",
              "layers.1.blocks.0.attn.relative_pos_value_table", "layers.2.blocks.5.mlp.fc2.weight", "layers.0.downsample.norm.weight",
              "layers.2.downsample.linear.weight", "upsamples.0.linear1.1.weight", "upsamples.2.linear2.0.bias", "classifier.1.running_var",
              "classifier.3.bias"]:
        assert k in keys, k
    assert not any(k.startswith("layers.3.downsample") for k in keys)
    assert m.layers[0].blocks[0].attn.relative_pos_query_table.shape == (64, 3, 16, 3)
    # stem_transformer=False variant: KPConv residual block + a TransitionDown in front of the transformer layers
    cfg = dict(helpers.ST_CFG, stem_transformer=False)
    m2 = MODELS.build(dict(type="ST-v1m1", drop_path_rate=0.0, **cfg))
    assert "stem_layer.1.unary_1.0.weight" in m2.state_dict() and "downsample.linear.weight" in m2.state_dict() and len(m2.layers) == 3


def _partition_case(n=600, seed=0, parity=0):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([1.5, 1.2, 0.4])
    batch = (torch.arange(n) >= (7 * n) // 12).long()
    ws = torch.tensor([0.3] * 3)
    kf, kc, wk = stratified.window_keys(xyz, batch, ws, xyz.min(0).values, parity)
    ds = torch.arange(0, n, 8)
    return xyz, batch, kf, kc, wk, ds


@pytest.mark.parametrize("parity", [0, 1])
def test_window_partition_properties(parity):
    """The oracle's restatement of grid_sample / get_indice_pairs (oracle/window_tables.py) on the product's per-point keys: every point is
    listed once in its window; the edge list contains every ordered pair of points that share a fine window, and every extra edge ends in
    an FPS-selected key of the same coarse window but another fine window."""
    from oracle import window_tables

    xyz, batch, kf, kc, wk, ds = _partition_case(parity=parity)
    p2v, counts = window_tables.p2v_from_keys(kf)
    listed = torch.cat([p2v[i, :counts[i]] for i in range(p2v.shape[0])])
    assert sorted(listed.tolist()) == list(range(600)) and int(counts.sum()) == 600
    for i in range(0, p2v.shape[0], 7):
        members = p2v[i, :counts[i]]
        assert len(set(kf[members].tolist())) == 1 and len(set(batch[members].tolist())) == 1
    p2v2, counts2 = window_tables.p2v_from_keys(kc)
    i0, i1 = window_tables.get_indice_pairs(p2v, counts, p2v2, counts2, ds, 600, wk)
    same_fine = kf[i0] == kf[i1]
    assert int(same_fine.sum()) == int((counts * counts).sum())
    extra = ~same_fine
    assert extra.any() and torch.isin(i1[extra], ds).all() and (batch[i0[extra]] == batch[i1[extra]]).all()
    assert (kc[i0[extra]] == kc[i1[extra]]).all() and (wk[i0[extra]] != wk[i1[extra]]).all()


@pytest.mark.parametrize("parity", [0, 1])
def test_window_rows_are_what_the_reference_construction_sorts_into(parity):
    """The per-query statement the device builder implements (csrc/window_edges.hip): after the reference's pair expansion and its
    stable sort by query, a query's row is [its fine window, ascending] ++ [the downsampled points of its coarse window that lie in
    another fine window, ascending] -- checked here against the oracle's tables by brute force."""
    from oracle import window_tables

    xyz, batch, kf, kc, wk, ds = _partition_case(n=500, seed=3, parity=parity)
    i0, i1, off, n_max, rel, flag = window_tables.window_edges(xyz, kf, kc, wk, ds, 0.6, 0.02, 59)
    isds = torch.zeros(500, dtype=torch.bool)
    isds[ds] = True
    ids = torch.arange(500)
    rows = 0
    for q in range(500):
        fine = ids[kf == kf[q]]
        coarse = ids[(kc == kc[q]) & isds & (wk != wk[q])]
        want = torch.cat([fine, coarse]).int()
        got = i1[off[q]:off[q + 1]]
        assert torch.equal(got, want), q
        assert (i0[off[q]:off[q + 1]] == q).all()
        rows = max(rows, want.shape[0])
    assert n_max == rows and int(off[-1]) == i0.shape[0] and int(flag) == 0
